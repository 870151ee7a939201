/*
 * ffcnn_hip.h -- additive, device-resident / batched C-ABI of libffcnn_hip.so.
 *
 * ffcnn.h and conv.h keep the reference's host-pointer interfaces (one frame,
 * H2D + D2H around every call).  Nothing in the reference has a batch
 * dimension or a device boundary, so the entry points below
 * are new names; each one states which reference code path it is the batched
 * counterpart of.  Plain C types only: pointers, sizes, ints.
 *
 * Device tensor layout ("CNHW"): a tensor of C channels for a batch of N frames
 * is C*N contiguous planes of H*W fp32, plane (c, n) at ((c*N + n)*H*W).  For
 * N == 1 this is exactly the reference's planar CHW tensor (ffcnn.c:436).  The
 * batch INPUT is the exception: frames are handed over frame-major, N x C x H x W
 * (each frame is one reference input tensor), and the first layer reads that.
 *
 * Error convention: functions returning int give 0 on success, negative on
 * failure; ffgpu_last_error() describes the last failure on the calling thread.
 * There is no CPU fallback anywhere: without a HIP device every call fails.
 */
#ifndef FFCNN_AMD_FFCNN_HIP_H
#define FFCNN_AMD_FFCNN_HIP_H

#include <stddef.h>
#include "ffcnn.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FFGPU_MAX_DET   128   /* boxes per frame in the fixed-size RECORD (the gather unit); a frame with more keeps   */
                              /* all of them in the executor's full list: ffgpu_exec_read_boxes                        */

/* Candidate capacity: the reference appends candidates to a buffer of net->bbox_max = 51 200 entries (ffcnn.c:243,463).
 * Here every anchor of every head cell owns a slot (3 * cells summed over the heads: 1 500 per frame at 320x320), so the
 * decode never drops one; if a frame ever has more than net->bbox_max candidates, the FIRST bbox_max in the reference's
 * emission order go into NMS, as in the reference.  ffgpu_exec_cand_capacity() returns the slots per frame. */

/* Per-frame detection record as it lies in device (and gathered host) memory.
 * This is the unit the multi-GPU gather moves: fixed size, 16 + 128*24 bytes. */
typedef struct {
    int  count;               /* boxes valid in box[] (post-NMS, source-image coords): min(nfull, FFGPU_MAX_DET)     */
    int  ncand;               /* candidates that passed ignore_thresh before NMS                                      */
    int  overflow;            /* bit 0: ncand > bbox_max (truncated like the reference); bit 2: nfull > FFGPU_MAX_DET  */
                              /* (bit 1 is set by ffgpu_pack_records on frames that lost boxes to its cap)             */
    int  nfull;               /* boxes that survived NMS (all of them are in the executor's full list)                 */
    BBOX box[FFGPU_MAX_DET];
} ffgpu_frame_dets;

typedef struct ffgpu_exec ffgpu_exec;     /* a planned executor: one NET x one batch size */

/* executor flags */
#define FFGPU_KEEP_ALL   1    /* no arena reuse: every tensor that is materialised can be read */
                              /* back (combine with FFGPU_NO_FUSE for one tensor per layer)    */
#define FFGPU_COMPAT_V6  2    /* reproduce conv-v6.c:422-441 (5x5 depthwise row oh-2 defect) */
#define FFGPU_NO_GRAPH   4    /* launch kernels eagerly instead of replaying a HIP graph     */
#define FFGPU_NO_FUSE    8    /* one kernel per reference layer (no cross-layer fusion)      */
#define FFGPU_HOST_DETS  16   /* the NMS kernel also writes the records to a pinned host     */
                              /* mirror (ffgpu_exec_dets_host): no D2H copy after a forward   */
#define FFGPU_SPLIT2     32   /* even batch: the two halves run as two parallel branches of   */
                              /* one graph (same results; fills the latency gaps of the       */
                              /* small-plane launches); not with FFGPU_KEEP_ALL               */

#define FFGPU_CONCURRENT 64   /* several executors of this device run at the same time (one   */
                              /* stream each): plan for throughput -- the small planes' tiles  */
                              /* are split over fewer waves (less redundant work per wave; a   */
                              /* lone chain is ~4 % slower, four in flight ~3.5 % faster)      */

#define FFGPU_BF16_PW   128   /* OPT-IN reduced precision: compute-bound pointwise layers (ic, oc >= 128, >= 16 384 pixels) round  */
                              /* inputs and weights to bf16 and accumulate in fp32 on the bf16 matrix cores (16x the fp32 rate). */
                              /* Never the default: outputs then differ from the reference by up to 2^-7 scale' sum|w x| per   */
                              /* value (env FFGPU_BF16_PW=1 sets it for every executor).  yolo-fastest has no such layer.      */

/* ---- process / device --------------------------------------------------- */
int         ffgpu_device_count(void);
int         ffgpu_set_device(int ordinal);            /* hipSetDevice for this thread        */
const char *ffgpu_last_error(void);
const char *ffgpu_build_info(void);                   /* "gfx950 ... <git describe/date>"    */

/* ---- weights (counterpart of ffcnn.c:211-239's weight_buf, in HBM) ------ */
/* Device address and byte size of the folded filter rows (same layout as
 * NET.weight_buf).  This is the buffer a multi-GPU job broadcasts (RCCL). */
int ffgpu_net_weights_dev(NET *net, void **dev_ptr, size_t *bytes);
/* Call after writing the device weights from outside (e.g. after a broadcast):
 * refreshes derived device-side packings.  Runs on `stream` (hipStream_t or NULL). */
int ffgpu_net_weights_commit(NET *net, void *stream);

/* ---- batched forward (counterpart of net_forward, ffcnn.c:476-520) ------ */
ffgpu_exec *ffgpu_exec_create(NET *net, int batch, int flags);
void        ffgpu_exec_destroy(ffgpu_exec *ex);
int         ffgpu_exec_batch(const ffgpu_exec *ex);
size_t      ffgpu_exec_arena_bytes(const ffgpu_exec *ex);
int         ffgpu_exec_kernel_count(const ffgpu_exec *ex);   /* launches per forward */
/* What one forward of this plan must move and compute: bytes = per launch the tensors it reads / writes once each
 * (fused launches keep their inner tensors on chip), flops = 2 x multiply-adds of every conv layer (ffcnn.c:374-379). */
int         ffgpu_exec_work_model(const ffgpu_exec *ex, double *hbm_bytes, double *flops);

/* Box rescale ratio for every frame of the batch (what net_input derives per
 * image, ffcnn.c:267-273).  Default s1 = s2 = 1 (boxes in network pixels). */
int ffgpu_exec_set_scale(ffgpu_exec *ex, int s1, int s2);

/* d_frames: device pointer, batch x C x H x W fp32 (frame-major).  Enqueues the
 * whole net + YOLO decode + NMS on `stream` (hipStream_t; NULL = the executor's
 * own stream) and returns without synchronising.  The executor replays ONE HIP
 * graph whatever buffer d_frames points to (the pointer and the box scale travel
 * through a small device parameter block written in stream order in front of
 * the graph): handing over a fresh buffer per call costs nothing extra. */
int ffgpu_exec_forward_dev(ffgpu_exec *ex, const float *d_frames, void *stream);

/* Same, from host memory (H2D copy included), then waits for completion. */
int ffgpu_exec_forward_host(ffgpu_exec *ex, const float *h_frames);

/* u8 BGR frames on the device (batch x h x ALIGN(3w,4) bytes, all the same size)
 * -> letterboxed fp32 input + forward: the batched net_input (ffcnn.c:259-289)
 * fused in front of the net.  Sets the scale from (w,h) like net_input does. */
int ffgpu_exec_forward_bgr_dev(ffgpu_exec *ex, const unsigned char *d_bgr, int w, int h,
                               const float mean[3], const float norm[3], void *stream);

/* Device address of the batch's ffgpu_frame_dets[batch] (valid after the
 * forward enqueued on the same stream completes). */
int ffgpu_exec_dets_dev(ffgpu_exec *ex, void **dev_ptr, size_t *bytes);
/* FFGPU_HOST_DETS executors: the pinned host mirror of the same `batch` records (valid once the
 * forward's stream has been synchronised); NULL + error otherwise. */
const ffgpu_frame_dets *ffgpu_exec_dets_host(ffgpu_exec *ex);
/* Record ring for the multi-GPU gather: from now on forward number k (k = 0, 1, ... counted on the device) ALSO
 * writes its `batch` records into slot k % slots of the caller-owned device buffer `dev_ring` (slots x batch
 * records), so groups of steps travel in one collective with no copy between graph launches.  Calling it again
 * (or with NULL) restarts the count / detaches.  Synchronises the executor's stream. */
int ffgpu_exec_set_ring(ffgpu_exec *ex, void *dev_ring, int slots);
/* The same with slot k at dev_ring + k * slot_records records (slot_records >= batch): E executors that take turns on E
 * streams share one ring when executor e gets dev_ring + e * batch records, slots / E slots and slot_records = E * batch
 * -- its forward k then lands in the ring's slot k * E + e, i.e. the global step number. */
int ffgpu_exec_set_ring_strided(ffgpu_exec *ex, void *dev_ring, int slots, int slot_records);
/* Synchronise the executor's last stream and copy the records to the host. */
int ffgpu_exec_read_dets(ffgpu_exec *ex, ffgpu_frame_dets *host_out, int max_frames);
/* ALL boxes of `frame` that survived NMS, score order, source-image coordinates (what the reference leaves in
 * net->bbox_list[0..bbox_num)): copies min(nfull, cap) of them and returns nfull.  Synchronises like read_dets. */
int ffgpu_exec_read_boxes(ffgpu_exec *ex, int frame, BBOX *host_out, int cap);
int ffgpu_exec_cand_capacity(const ffgpu_exec *ex);     /* candidate slots per frame (see above)                       */
int ffgpu_exec_graph_captures(const ffgpu_exec *ex);    /* HIP graphs captured so far by this executor (1 after any    */
                                                        /* number of forwards on any number of buffers)                */

/* FFGPU_KEEP_ALL executors only: copy layer `layer`'s OUTPUT for frame `frame`
 * into host_out (oc*oh*ow floats, reference CHW order).  layer == -1 gives the
 * network input as the first layer saw it.  Pre-NMS candidates: layer == -2
 * writes the frame's candidates (up to ffgpu_exec_cand_capacity() BBOX, in network
 * pixels, reference emission order) and returns their count. */
int ffgpu_exec_read_layer(ffgpu_exec *ex, int layer, int frame, float *host_out, size_t cap_floats);

/* FFGPU_KEEP_ALL executors only: host_out[i] = a 64-bit position-dependent hash of
 * layer i's output over the WHOLE batch (every bit of every frame; computed on
 * the device behind the last forward), 0 for layers this executor does not
 * materialise.  cap >= NET.layer_num.  Returns the number of layers hashed.
 * Two forwards of the same frames must give the same values: the reproducibility
 * watch of the concurrency tests reads 8 bytes per layer instead of activations. */
int ffgpu_exec_hash_layers(ffgpu_exec *ex, unsigned long long *host_out, int cap);

/* Mean device time per layer KIND over the last profiled forward, in micro-
 * seconds, indexed by LAYER_TYPE_* (counterpart of ENABLE_NET_PROFILE,
 * ffcnn.c:33,494-510).  Runs one eager forward with hipEvents around each step. */
int ffgpu_exec_profile(ffgpu_exec *ex, const float *d_frames, float us_by_kind[LAYER_TYPE_TOTOAL]);

/* Per-launch breakdown of one eager forward: for step i, layer_of[i] is the
 * reference layer index it implements (-1: executor bookkeeping), us[i] its
 * device time.  Returns the number of steps (<= cap) or a negative error. */
int ffgpu_exec_profile_steps(ffgpu_exec *ex, const float *d_frames, int *layer_of, float *us, int cap);
/* The HBM byte model of ffgpu_exec_work_model step by step: hbm_bytes[i] = what step i must move (its input and
 * output tensors and filter rows, each once), layer_of[i] as above.  Returns the number of steps (<= cap). */
int ffgpu_exec_step_model(const ffgpu_exec *ex, int *layer_of, double *hbm_bytes, int cap);

/* ---- all GPUs of one node from one C process (counterpart of calling net_forward once per frame) ------
 * A batch of `global_batch` independent frames is cut into contiguous shards, one per device (ffgpu_shard_range); every
 * device holds its own copy of the folded filter rows -- broadcast ONCE from rank 0 over RCCL (ncclBroadcast) inside
 * ffgpu_node_create -- and runs the whole net on its shard (own executor, stream and HIP graph); per forward the shards'
 * detection records are gathered on rank 0 (grouped ncclSend / ncclRecv over xGMI) and land in the caller's host buffer
 * in global frame order.  No all-reduce, no activation exchange.  RCCL is loaded at run time (librccl.so.1) the first
 * time a node is created without FFGPU_NODE_LOOPBACK; single-GPU users never load it. */
typedef struct ffgpu_node ffgpu_node;
#define FFGPU_NODE_LOOPBACK 1   /* node_flags: peer copies instead of RCCL; several ranks may share a device (tests) */
#define FFGPU_NODE_DEPTH(n) (((n) & 0xf) << 8)   /* node_flags: up to n (1..8) steps in flight: slot = step % n has its own executor, */
                                /* compute stream and input buffer per device and its own gather / host buffers                   */

/* frames [lo, hi) of `rank` out of `world`: sizes differ by at most one, earlier ranks take the extra */
void        ffgpu_shard_range(int total, int rank, int world, int *lo, int *hi);
/* devices: ndev device ordinals (rank 0 = the device `net` was loaded on) or NULL for that device and the next ndev - 1.
 * exec_flags: FFGPU_* executor flags for the per-device executors (FFGPU_COMPAT_V6, FFGPU_NO_GRAPH, FFGPU_SPLIT2 ...). */
ffgpu_node *ffgpu_node_create(NET *net, int ndev, const int *devices, int global_batch, int exec_flags, int node_flags);
void        ffgpu_node_destroy(ffgpu_node *node);
int         ffgpu_node_ndev(const ffgpu_node *node);
int         ffgpu_node_shard(const ffgpu_node *node, int rank, int *lo, int *hi, int *device);
int         ffgpu_node_set_scale(ffgpu_node *node, int s1, int s2);       /* as ffgpu_exec_set_scale, every rank */
int         ffgpu_node_depth(const ffgpu_node *node);
/* communicators ncclCommInitAll created for this node: ndev on the RCCL path, 0 with one device or FFGPU_NODE_LOOPBACK */
int         ffgpu_node_rccl_ranks(const ffgpu_node *node);
/* rank's input shard on ITS device, (hi - lo) x C x H x W fp32 frame-major, owned by the node (the buffer the NEXT
 * submitted step reads; ffgpu_node_input_slot_dev names a slot explicitly): fill it there ... */
float      *ffgpu_node_input_dev(ffgpu_node *node, int rank);
float      *ffgpu_node_input_slot_dev(ffgpu_node *node, int rank, int slot);
/* ... then run: forward on every device, gather, records of all global_batch frames to host_out; returns when they are there */
int         ffgpu_node_forward(ffgpu_node *node, ffgpu_frame_dets *host_out);
/* or hand over the whole batch in host memory (global_batch x C x H x W fp32): shards are copied to their devices first */
int         ffgpu_node_forward_host(ffgpu_node *node, const float *h_frames, ffgpu_frame_dets *host_out);
/* Pipelined form: submit enqueues the next step on every device (h_frames may be NULL: the slot's input buffers are used as
 * they are) plus its gather and returns the step's ticket (>= 0) without waiting; wait(ticket) blocks until that step's
 * records are on the host and copies them.  At most `depth` tickets may be outstanding.  h_frames is copied into the slot's
 * page-locked staging buffer before submit returns: the caller's buffer is free again at once -- with ONE exception: frames that
 * lie inside memory from ffgpu_host_alloc (page-locked memory this library owns) are uploaded straight from there by an
 * asynchronous DMA and must stay untouched until ffgpu_node_wait(ticket) has returned. */
long        ffgpu_node_submit(ffgpu_node *node, const float *h_frames);
int         ffgpu_node_wait(ffgpu_node *node, long ticket, ffgpu_frame_dets *host_out);
/* The loop above as one call: `steps` steps from the slots' input buffers, `depth` of them in flight (collect step i - depth,
 * submit step i), every step's records brought to the host; host_out (may be NULL) receives the last step's.  If a step fails the
 * steps still in flight are drained (their records dropped) before the error is returned: the node stays usable. */
int         ffgpu_node_run(ffgpu_node *node, long steps, ffgpu_frame_dets *host_out);

/* ---- single operators on device tensors (CNHW, any batch) --------------- */
/* Counterpart of groupconv (conv.h:4-7) without the host round trip.  d_in is
 * ic*batch planes of ih*iw, d_out oc*batch planes of oh*ow; d_filt as conv.h.
 * variant: 0 = auto (what the executor would pick), otherwise a specific kernel
 * id (FFGPU_K_*) for testing/benchmarking; unsupported combinations fail. */
int ffgpu_groupconv_dev(const float *d_in, const float *d_filt, float *d_out, int batch,
                        int iw, int ih, int ic, int groups, int pad, int stride,
                        int fs, int fn, int ow, int oh, int oc, int act,
                        int flags, int variant, void *stream);

enum {
    FFGPU_K_AUTO = 0,
    FFGPU_K_GENERIC = 1,      /* any fs/stride/pad/groups: one thread per output          */
    FFGPU_K_DW_STREAM = 2,    /* depthwise 3x3 s1: register sliding window, 16 B loads    */
    FFGPU_K_DW_LDS = 3,       /* depthwise 3x3/5x5, s1/s2: whole planes staged in LDS     */
    FFGPU_K_PW_MFMA = 4,      /* 1x1: fp32 MFMA 16x16x4, streaming (bandwidth-bound)      */
    FFGPU_K_PW_GEMM = 5,      /* 1x1: fp32 MFMA GEMM tile for compute-bound shapes (ic, oc >= 128) */
    FFGPU_K_DENSE_SMALL = 7,  /* dense 3x3/5x5 with <= 8 input channels (the first layer)   */
    FFGPU_K_IGEMM = 8,        /* dense KxK, groups == 1: implicit GEMM on fp32 MFMA (im2col gathered on the fly) */
    FFGPU_K_PW_BF16 = 9,      /* 1x1, opt-in (FFGPU_BF16_PW): bf16 inputs, fp32 accumulation, v_mfma_f32_32x32x16_bf16 */
    FFGPU_K_GROUP_THIN = 10,  /* 2..7 input channels per group (grouped or dense), any fs / stride / pad: scalar filter taps, several outputs per lane */
    FFGPU_K_PW_X3 = 11,       /* 1x1: fp32-equivalent results from SPLIT operands (three exact bf16 parts each, six partial products, fp32 accumulation) on v_mfma_f32_16x16x32_bf16 */
    FFGPU_K_CONV_X3 = 12,     /* dense 3x3 / stride 1 / pad 1, one group: the same split-operand arithmetic as FFGPU_K_PW_X3 (ffgpu_conv_x3.inc) */
    FFGPU_K_PW_X3T = 13       /* 1x1: the same split-operand arithmetic as a tiled, double-buffered GEMM (ffgpu_pw_x3t.inc): every input value split once per 256 output channels */
};

/* name of the kernel `variant` resolves to for this shape (for logs/benches) */
const char *ffgpu_groupconv_kernel_name(int batch, int iw, int ih, int ic, int groups, int pad,
                                        int stride, int fs, int fn, int variant);

/* Times `iters` launches of one conv on `stream` with hipEvents recorded on that
 * stream (after `warmup` untimed launches); returns mean microseconds per launch
 * or a negative value on error. */
float ffgpu_groupconv_time_dev(const float *d_in, const float *d_filt, float *d_out, int batch,
                               int iw, int ih, int ic, int groups, int pad, int stride,
                               int fs, int fn, int ow, int oh, int oc, int act,
                               int flags, int variant, int warmup, int iters, void *stream);

/* Fused block: 1x1 expand -> depthwise 3x3 (stride 1|2, pad 1) -> 1x1 project [+ residual], i.e.
 * three consecutive groupconv calls of the reference plus the shortcut that follows them
 * (ffcnn.c:418-423) in one kernel; the expanded tensors never leave the CU.  CNHW device
 * tensors; d_w1/d_wd/d_w2 are the three layers' filter rows (conv.h layout); d_res may be NULL.
 * iters > 0: returns mean microseconds per launch (HIP events on `stream`) instead of 0. */
float ffgpu_irb_dev(const float *d_in, const float *d_w1, const float *d_wd, const float *d_w2,
                    const float *d_res, float *d_out, int batch, int iw, int ih, int ic, int ec, int oc,
                    int stride, int act1, int actd, int act2, int res_act, int warmup, int iters, void *stream);

/* Fused pair: depthwise K x K (K = 3 | 5, stride 1, pad K / 2) -> pointwise 1x1, i.e. two consecutive groupconv calls of the
 * reference in one kernel (the depthwise tensor never leaves the CU).  CNHW device tensors; d_wd / d_wp are the two layers'
 * filter rows (conv.h layout).  iters > 0: returns mean microseconds per launch (HIP events on `stream`) instead of 0. */
float ffgpu_dwpw_dev(const float *d_in, const float *d_wd, const float *d_wp, float *d_out, int batch, int iw, int ih,
                     int c, int oc, int fs, int actd, int actp, int warmup, int iters, void *stream);

/* ---- compact records for the multi-GPU gather ("fixed-size detection records to rank 0"): the
 * `batch` records of each of `nslots` steps (slot s starts at record s * slot_stride_records of d_records, e.g. a ring
 * set with ffgpu_exec_set_ring) are packed into nslots blocks of ffgpu_packed_records_bytes(batch, cap) bytes:
 *     int total, over, batch, cap | { int count, ncand, overflow, nfull } x batch | BBOX box[cap]
 * boxes of the frames behind each other in frame order (a frame's first box sits at the sum of the counts before it).  A step with more than `cap` boxes keeps the first `cap`
 * (over = 1, overflow |= 2 on the frames that lost boxes).  ffcnn_amd/dist.py unpacks them on the host. */
size_t ffgpu_packed_records_bytes(int batch, int cap);
/* host side: one packed block -> `batch` fixed-size records (unused box slots zero, as the NMS kernel leaves them).  Returns 0, or 1
 * if the block had dropped boxes / does not describe (batch, cap) -- the caller then fetches the full-size records --, -1 on bad arguments.
 * Pure host code (what ffgpu_node_wait runs on the gathered blocks). */
int    ffgpu_unpack_records(const void *block, int batch, int cap, ffgpu_frame_dets *out);
int    ffgpu_pack_records(const void *d_records, int nslots, long slot_stride_records, int batch, int cap, void *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FFCNN_AMD_FFCNN_HIP_H */
