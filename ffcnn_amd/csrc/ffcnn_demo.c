/*
 * ffcnn_demo.c -- command-line harness with the behaviour of the reference's demo main
 * (ffcnn.c:552-593, built there with -D_TEST_):  ffcnn_hip_demo [n] [bmp] [cfg] [weights]
 * loads a 24-bit BMP, builds the net at the image's geometry (rounded up to 32), runs n times
 * net_input + net_forward on the GPU, prints the detections in the reference's format, outlines
 * them in green and writes out.bmp.  Host harness only: plain C on top
 * of the ffcnn.h API of libffcnn_hip.so.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ffcnn.h"

typedef struct { int w, h, pitch; unsigned char *px; } image_t;       /* top-down BGR rows */

static int image_load(image_t *im, const char *path)                   /* counterpart of bmpfile.c:42-69 */
{
    unsigned char hdr[54];
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    if (fread(hdr, 1, sizeof hdr, fp) != sizeof hdr) { fclose(fp); return -1; }
    uint32_t w, h;
    memcpy(&w, hdr + 18, 4); memcpy(&h, hdr + 22, 4);
    im->w = (int)w; im->h = (int)h; im->pitch = ((int)w * 3 + 3) & ~3;
    im->px = (unsigned char *)malloc((size_t)im->pitch * im->h);
    if (im->px)
        for (int y = im->h - 1; y >= 0; y--)                           /* file rows are bottom-up */
            if (fread(im->px + (size_t)y * im->pitch, (size_t)im->pitch, 1, fp) != 1) break;
    fclose(fp);
    return im->px ? 0 : -1;
}

static int image_save(const image_t *im, const char *path)             /* bmpfile.c:78-106 */
{
    unsigned char hdr[54] = { 'B', 'M' };
    const uint32_t bytes = (uint32_t)im->pitch * (uint32_t)im->h, total = bytes + 54, off = 54, dib = 40;
    const uint32_t w = (uint32_t)im->w, h = (uint32_t)im->h;
    const uint16_t planes = 1, bpp = 24;
    memcpy(hdr + 2, &total, 4); memcpy(hdr + 10, &off, 4); memcpy(hdr + 14, &dib, 4);
    memcpy(hdr + 18, &w, 4); memcpy(hdr + 22, &h, 4); memcpy(hdr + 26, &planes, 2); memcpy(hdr + 28, &bpp, 2);
    memcpy(hdr + 34, &bytes, 4);
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    fwrite(hdr, 1, sizeof hdr, fp);
    for (int y = im->h - 1; y >= 0; y--) fwrite(im->px + (size_t)y * im->pitch, (size_t)im->pitch, 1, fp);
    fclose(fp);
    return 0;
}

static void put(image_t *im, int x, int y, int r, int g, int b)
{
    if (x < 0 || y < 0 || x >= im->w || y >= im->h) return;
    unsigned char *p = im->px + (size_t)y * im->pitch + x * 3;
    p[0] = (unsigned char)b; p[1] = (unsigned char)g; p[2] = (unsigned char)r;
}

static void outline(image_t *im, int x1, int y1, int x2, int y2, int r, int g, int b)   /* bmpfile.c:146-157 */
{
    for (int x = x1; x <= x2; x++) { put(im, x, y1, r, g, b); put(im, x, y2, r, g, b); }
    for (int y = y1; y <= y2; y++) { put(im, x1, y, r, g, b); put(im, x2, y, r, g, b); }
}

static int now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int)(ts.tv_sec * 1000 + ts.tv_nsec / 1000000);
}

void net_profile(NET *net);

int main(int argc, char **argv)
{
    float mean[3] = { 0.f, 0.f, 0.f }, norm[3] = { 1 / 255.f, 1 / 255.f, 1 / 255.f };
    int n = argc > 1 ? atoi(argv[1]) : 10;
    char *bmp = argc > 2 ? argv[2] : "test.bmp";
    char *cfg = argc > 3 ? argv[3] : "yolo-fastest-1.1.cfg";
    char *wts = argc > 4 ? argv[4] : "yolo-fastest-1.1.weights";
    image_t im = { 0 };
    printf("file_bmp    : %s\n", bmp);
    printf("file_cfg    : %s\n", cfg);
    printf("file_weights: %s\n", wts);
    if (image_load(&im, bmp) != 0) { printf("failed to load bmp file: %s !\n", bmp); return -1; }
    NET *net = net_load(cfg, wts, im.w, im.h);
    if (!net) { printf("net_load failed (cfg unreadable or no HIP device)\n"); free(im.px); return -1; }
    net_dump(net);
    int t0 = now_ms();
    for (int i = 0; i < n; i++) {
        net_input(net, im.px, im.w, im.h, mean, norm);
        net_forward(net);
    }
    printf("%d times inference: %d ms\n", n, now_ms() - t0);
    net_profile(net);
    for (int i = 0; i < net->bbox_num; i++) {
        const BBOX *b = net->bbox_list + i;
        printf("score: %.2f, category: %2d, rect: (%3d %3d %3d %3d)\n", b->score, b->type, (int)b->x1, (int)b->y1, (int)b->x2, (int)b->y2);
        outline(&im, (int)b->x1, (int)b->y1, (int)b->x2, (int)b->y2, 0, 255, 0);
    }
    net_free(net);
    image_save(&im, "out.bmp");
    free(im.px);
    return 0;
}
