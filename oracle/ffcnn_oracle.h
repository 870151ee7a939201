/*
 * ffcnn_oracle.h -- CPU restatement of the ffcnn forward path.  TEST INFRASTRUCTURE.
 *
 * This is the checker the HIP path is compared against, not a product code
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  libffcnn_hip.so never links or calls anything in oracle/.
 *
 * Pinning: the oracle is validated (tests/test_oracle_vs_ref.py, run where
 * /root/reference exists) against the reference itself compiled unmodified
 * into oracle/_ref/ (see oracle/Makefile), and (everywhere) against the golden
 * vectors in tests/golden/ that oracle/gen_golden.py produced from that build.
 * Every function cites the reference file:line whose arithmetic it restates.
 *
 * Deliberate differences from the reference implementation:
 *  - every activation is kept (one buffer per layer) so per-layer parity can
 *    be read back; the reference frees tensors as their refcount drops;
 *  - boxes live in their own array instead of aliasing the input tensor;
 *  - the cfg is parsed line by line with exact key matching.
 */
#ifndef FFCNN_ORACLE_H
#define FFCNN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_CONV, ORC_AVGPOOL, ORC_MAXPOOL, ORC_UPSAMPLE, ORC_DROPOUT,
       ORC_SHORTCUT, ORC_ROUTE, ORC_YOLO };

typedef struct {
    int    kind;
    int    iw, ih, ic;            /* input geometry                         */
    int    ow, oh, oc;            /* output geometry                        */
    int    fs, fn, stride, pad, groups, batchnorm, act;
    int    ndep, dep[4];
    int    classes, anchors[3][2];
    float  thresh, scale_xy;
    float *filt;                  /* -> into orc_net.weights                */
    float *out;                   /* this layer's OUTPUT activations        */
} orc_layer;

typedef struct { int type; float score, x1, y1, x2, y2; } orc_box;

typedef struct {
    int        nlayers;
    orc_layer *layers;
    int        in_w, in_h, in_c;
    float     *input;             /* in_c x in_h x in_w                     */
    int        nweights;          /* padded float count                     */
    float     *weights;
    orc_box   *boxes;             /* NMS'd, source-image coordinates        */
    int        nboxes;
    orc_box   *cand;              /* pre-NMS candidates of the last forward */
    int        ncand, cap;
    int        s1, s2;
    int        weights_consumed;  /* floats read from the .weights file     */
} orc_net;

/* ---- single operators (host pointers, one frame) ---------------------- */

/* compat_v6 != 0 reproduces conv-v6.c:422-441 (5x5 depthwise, row oh-2 drops
 * tap row 0); 0 is the textbook semantics of conv-v0.c..conv-v5.c. */
void orc_groupconv(const float *in, const float *filt, float *out,
                   int iw, int ih, int ic, int groups, int pad, int stride,
                   int fs, int fn, int ow, int oh, int oc, int act, int compat_v6);
void orc_pool(const float *in, float *out, int w, int h, int c, int fs, int stride, int is_max);
void orc_upsample(const float *in, float *out, int w, int h, int c, int stride);
void orc_shortcut(const float *a, const float *b, float *out, int n, int act);
float orc_activate(float x, int act);

/* decode one yolo head; appends to cand[*ncand..cap). */
void orc_yolo(const float *in, int w, int h, int classes, const int anchors[3][2],
              float thresh, float scale_xy, int netw, int neth,
              orc_box *cand, int *ncand, int cap);
/* in-place sort + greedy class-aware NMS + rescale; returns survivors. */
int  orc_nms(orc_box *b, int n, float thresh, int use_min, int s1, int s2);

/* ---- whole net -------------------------------------------------------- */
orc_net *orc_load(const char *cfg, const char *weights, int inputw, int inputh);
void     orc_free(orc_net *n);
void     orc_input(orc_net *n, const unsigned char *bgr, int w, int h,
                   const float mean[3], const float norm[3]);
void     orc_forward(orc_net *n, int compat_v6);
/* output tensor of layer i (i in [0,nlayers)); NULL for yolo layers. */
const float *orc_layer_out(const orc_net *n, int i);
int      orc_dump(const orc_net *n, char *buf, int buflen);  /* net_dump text */

/* minimal 24-bit BMP reader (top-down rows, stride ALIGN(3w,4)); returns
 * malloc'd pixels or NULL. Restates bmpfile.c:42-69. */
unsigned char *orc_bmp_load(const char *path, int *w, int *h);

#ifdef __cplusplus
}
#endif
#endif
