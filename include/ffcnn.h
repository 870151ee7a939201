/*
 * ffcnn.h -- net / layer ABI of the MI355X-native ffcnn forward path.
 *
 * This header is the Level-2 drop-in boundary: it keeps
 * the struct layout and the five entry points an application written against
 * rockcarry/ffcnn's ffcnn.h already uses, so such an application re-links
 * against libffcnn_hip.so without source changes.
 *
 *   reference interface replaced           here
 *   ------------------------------------   ------------------------------
 *   ffcnn.h:4-14   layer type enum         FFCNN layer kinds (same values)
 *   ffcnn.h:16-27  LAYER  (120 bytes)      LAYER  (same field order/sizes)
 *   ffcnn.h:29-32  BBOX   ( 24 bytes)      BBOX
 *   ffcnn.h:34-46  NET    (104 bytes)      NET
 *   ffcnn.h:48-52  net_load/free/input/forward/dump
 *
 * The struct sizes are checked at compile time below (x86-64 LP64).  Device
 * state (weights in HBM, activation arena, HIP graphs) lives in a private
 * block that net_load() places behind the LAYER array; nothing in the public
 * structs changes size or meaning.
 *
 * Tensor convention (same as the reference): layer_list[i] describes the
 * INPUT tensor of layer i (w,h,c,data) plus layer i's parameters; the output
 * shape of layer i is layer_list[i+1].{w,h,c}.  There are layer_num+1 entries.
 * Host tensors are planar fp32, index c*h*w + y*w + x.
 *
 * The batched / device-resident API (new, additive) is in ffcnn_hip.h.
 */
#ifndef FFCNN_AMD_FFCNN_H
#define FFCNN_AMD_FFCNN_H

#ifdef __cplusplus
extern "C" {
#endif

/* layer kinds; numeric values are part of the ABI (LAYER.type, NET.timeused[]) */
enum {
    LAYER_TYPE_CONV     = 0,
    LAYER_TYPE_AVGPOOL  = 1,
    LAYER_TYPE_MAXPOOL  = 2,
    LAYER_TYPE_UPSAMPLE = 3,
    LAYER_TYPE_DROPOUT  = 4,
    LAYER_TYPE_SHORTCUT = 5,
    LAYER_TYPE_ROUTE    = 6,
    LAYER_TYPE_YOLO     = 7,
    LAYER_TYPE_TOTOAL   = 8   /* (sic) spelling kept: it is the name callers know */
};

typedef struct {
    int    type;             /* LAYER_TYPE_*                                        */
    int    refcnt;           /* consumers still to run (liveness of .data)          */
    float *data;             /* host tensor of this layer's INPUT (may be NULL)     */
    float *filter;           /* conv: fn rows of ALIGN(fs*fs*c/groups,4)+4 floats:  */
                             /*   [taps..., 0-pad, scale', bias', mean, var]        */
    int    w, h, c;          /* input tensor geometry                               */
    int    pad, stride;      /* conv: pad = cfg pad ? fs/2 : 0                      */
    int    fn, fs, groups;   /* filter count (= output channels), size, groups      */
    int    batchnorm;        /* 1 when the cfg had batch_normalize=1 (already folded)*/
    int    activation;       /* 0 linear, 1 relu, 2 leaky(0.1), -1 unknown=linear   */
    int    depend_list[4];   /* shortcut: [from]; route: up to 4 source layer ids   */
    int    depend_num;

    int    class_num;        /* yolo head                                           */
    int    anchor_list[3][2];
    float  ignore_thres;
    float  scale_x_y;
} LAYER;

typedef struct {
    int   type;              /* class index                                         */
    float score;
    float x1, y1, x2, y2;    /* source-image pixels after net_forward               */
} BBOX;

typedef struct {
    LAYER *layer_list;       /* layer_num + 1 entries, same allocation as NET       */
    int    layer_num;
    BBOX  *bbox_list;        /* valid [0, bbox_num) after net_forward               */
    int    bbox_num;
    int    bbox_max;
    int    s1, s2;           /* box rescale ratio set by net_input (src / resized)  */
    int    weight_size;      /* floats in weight_buf (padded rows)                  */
    float *weight_buf;       /* host copy of every conv filter row, contiguous      */
    float *cnntempbuf;       /* conv scratch owned by the groupconv plug-in         */
    int    cnnbufsize;       /* ... its size in floats                              */
    int    timeused[LAYER_TYPE_TOTOAL]; /* per-kind accumulated ms (FFCNN_PROFILE=1)*/
} NET;

#if defined(__x86_64__) || defined(__aarch64__)
typedef char ffcnn_abi_check_layer[(sizeof(LAYER) == 120) ? 1 : -1];
typedef char ffcnn_abi_check_bbox [(sizeof(BBOX)  ==  24) ? 1 : -1];
typedef char ffcnn_abi_check_net  [(sizeof(NET)   == 104) ? 1 : -1];
#endif

/*
 * net_load: parse a darknet .cfg, load + BN-fold the .weights, upload them to
 * the GPU and plan the device executor.  inputw/inputh == 0 -> cfg geometry,
 * otherwise rounded up to a multiple of 32.  Returns NULL when the cfg cannot
 * be read, on allocation failure, or when no HIP device/extension is usable
 * (there is deliberately no CPU fallback in this library).  A missing
 * weights file is tolerated (all-zero filters), as in the reference.
 * Replaces: reference ffcnn.c:114-247.
 */
NET *net_load(char *cfg_path, char *weights_path, int inputw, int inputh);

/* Releases host and device resources. NULL-safe. Replaces ffcnn.c:249-257. */
void net_free(NET *net);

/*
 * net_input: BGR u8 image (row stride ALIGN(3*w,4)) -> planar RGB fp32 input
 * tensor, aspect-preserving nearest-neighbour resize into the top-left corner,
 * (byte - mean[ch]) * norm[ch].  Clears the previous frame's boxes and sets
 * s1/s2.  Replaces ffcnn.c:259-289.
 */
void net_input(NET *net, unsigned char *bgr, int w, int h, float *mean, float *norm);

/*
 * net_forward: one frame through the whole graph on the GPU (upload of
 * layer_list[0].data, every layer as a HIP kernel, YOLO decode + NMS on the
 * device), then boxes into bbox_list/bbox_num.  Replaces ffcnn.c:476-520.
 */
void net_forward(NET *net);

/* Prints the layer table in the reference's format. Replaces ffcnn.c:522-548. */
void net_dump(NET *net);

/* Prints timeused[]; present in the reference object but not in its header
 * (ffcnn.c:550).  Exported for link compatibility. */
void net_profile(NET *net);

#ifdef __cplusplus
}
#endif
#endif /* FFCNN_AMD_FFCNN_H */
