/*
 * ffcnn_node_demo.c -- a batch of frames on every GPU of the node from ONE plain-C process:
 *     ffcnn_node_demo [ndev] [global_batch] [steps] [bmp] [cfg] [weights]
 * net_load as the reference does (ffcnn.h), then the additive ffgpu_node_* calls of ffcnn_hip.h: the folded weights are
 * broadcast from GPU 0 to the others over RCCL, every step cuts `global_batch` frames (copies of the letterboxed image,
 * frame k shifted by k columns so the shards differ) into contiguous shards, runs the whole net on each GPU and gathers
 * the detection records on GPU 0.  Prints the boxes of frame 0 in the reference CLI's format (ffcnn.c:586) and the rate.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ffcnn.h"
#include "ffcnn_hip.h"

static unsigned char *bmp_load(const char *path, int *w, int *h)
{
    unsigned char hdr[54];
    FILE *fp = fopen(path, "rb");
    if (!fp || fread(hdr, 1, sizeof hdr, fp) != sizeof hdr) { if (fp) fclose(fp); return NULL; }
    uint32_t ww, hh;
    memcpy(&ww, hdr + 18, 4); memcpy(&hh, hdr + 22, 4);
    const size_t pitch = ((size_t)ww * 3 + 3) & ~(size_t)3;
    unsigned char *px = (unsigned char *)malloc(pitch * hh);
    for (int y = (int)hh - 1; px && y >= 0; y--)
        if (fread(px + (size_t)y * pitch, pitch, 1, fp) != 1) break;
    fclose(fp);
    *w = (int)ww; *h = (int)hh;
    return px;
}

int main(int argc, char **argv)
{
    const int want_dev = argc > 1 ? atoi(argv[1]) : 0, steps = argc > 3 ? atoi(argv[3]) : 50;
    const char *bmp = argc > 4 ? argv[4] : "test.bmp", *cfg = argc > 5 ? argv[5] : "yolo-fastest-1.1.cfg";
    const char *wts = argc > 6 ? argv[6] : "yolo-fastest-1.1.weights";
    const int ndev = want_dev > 0 ? want_dev : ffgpu_device_count();
    const int batch = argc > 2 && atoi(argv[2]) > 0 ? atoi(argv[2]) : 32 * ndev;
    float mean[3] = { 0.f, 0.f, 0.f }, norm[3] = { 1 / 255.f, 1 / 255.f, 1 / 255.f };
    int w = 0, h = 0;
    unsigned char *img = bmp_load(bmp, &w, &h);
    if (!img) { fprintf(stderr, "cannot read %s\n", bmp); return 1; }
    NET *net = net_load((char *)cfg, (char *)wts, 0, 0);
    if (!net) { fprintf(stderr, "net_load failed: %s\n", ffgpu_last_error()); return 1; }
    net_input(net, img, w, h, mean, norm);                           /* the letterboxed fp32 frame in layer_list[0].data */
    const LAYER *l0 = net->layer_list;
    const size_t plane = (size_t)l0->w * l0->h, fl = plane * l0->c;
    float *frames = (float *)malloc(sizeof(float) * fl * batch);
    ffgpu_frame_dets *dets = (ffgpu_frame_dets *)malloc(sizeof(ffgpu_frame_dets) * batch);
    if (!frames || !dets) return 1;
    for (int k = 0; k < batch; k++)                                  /* frame k = the image rolled by k columns */
        for (size_t row = 0; row < (size_t)l0->c * l0->h; row++)
            for (int x = 0; x < l0->w; x++)
                frames[k * fl + row * l0->w + (size_t)((x + k) % l0->w)] = l0->data[row * l0->w + x];
    int depth = getenv("FFCNN_NODE_DEPTH") ? atoi(getenv("FFCNN_NODE_DEPTH")) : 8;   /* steps in flight (1..8) */
    if (depth < 1) depth = 1;
    if (depth > 8) depth = 8;
    ffgpu_node *node = ffgpu_node_create(net, ndev, NULL, batch, FFGPU_CONCURRENT, FFGPU_NODE_DEPTH(depth));
    if (!node) { fprintf(stderr, "ffgpu_node_create failed: %s\n", ffgpu_last_error()); return 1; }
    ffgpu_node_set_scale(node, net->s1, net->s2);
    for (int r = 0; r < ndev; r++) {
        int lo, hi, dev;
        ffgpu_node_shard(node, r, &lo, &hi, &dev);
        printf("rank %d: device %d, frames [%d, %d)\n", r, dev, lo, hi);
    }
    for (int s = 0; s < depth; s++)                                  /* every slot's input buffers get the frames once */
        if (ffgpu_node_forward_host(node, frames, dets)) { fprintf(stderr, "forward failed: %s\n", ffgpu_last_error()); return 1; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    /* `depth` steps in flight: collect step i - depth, submit step i (ffgpu_node_submit / ffgpu_node_wait in a loop, as one call) */
    if (ffgpu_node_run(node, steps, dets)) { fprintf(stderr, "ffgpu_node_run failed: %s\n", ffgpu_last_error()); return 1; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    for (int i = 0; i < dets[0].count; i++) {
        const BBOX *b = dets[0].box + i;
        printf("score: %.2f, category: %2d, rect: (%3d %3d %3d %3d)\n", b->score, b->type, (int)b->x1, (int)b->y1, (int)b->x2, (int)b->y2);
    }
    int total = 0;
    for (int k = 0; k < batch; k++) total += dets[k].count;
    printf("%d GPUs, %d frames per step (inputs resident on the devices), %d steps: %.1f frames/s; %d boxes in the batch\n",
           ndev, batch, steps, steps > 0 ? (double)batch * steps / dt : 0.0, total);
    ffgpu_node_destroy(node);
    net_free(net);
    free(frames); free(dets); free(img);
    return 0;
}
