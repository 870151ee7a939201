/*
 * san_driver.c -- walks the host side of the ffcnn.h API (ffcnn_host.c) under AddressSanitizer + UBSan:
 *     ffcnn_host_san <cfg> <weights | -> [inputw inputh [imgw imgh]]
 * net_load -> net_dump -> net_input (a synthetic BGR image, pitch padded to 4 as bmpfile.c:37 hands it over) -> net_forward
 * (fails in the stub, must not crash) -> net_profile -> net_free.  Exit code 0 = the walk ended (whether or not net_load
 * accepted the files); a sanitizer report aborts with its own exit code (-fno-sanitize-recover).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ffcnn.h"
#include "ffcnn_hip.h"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s cfg weights|- [inputw inputh [imgw imgh]]\n", argv[0]); return 2; }
    const int iw = argc > 4 ? atoi(argv[3]) : 0, ih = argc > 4 ? atoi(argv[4]) : 0;
    const int w = argc > 6 ? atoi(argv[5]) : 37, h = argc > 6 ? atoi(argv[6]) : 23;
    NET *net = net_load(argv[1], strcmp(argv[2], "-") ? argv[2] : NULL, iw, ih);
    if (!net) { printf("net_load: NULL (%s)\n", ffgpu_last_error()); return 0; }
    printf("net_load: %d layers, weight_size %d, bbox_max %d\n", net->layer_num, net->weight_size, net->bbox_max);
    net_dump(net);
    const size_t pitch = ((size_t)w * 3 + 3) / 4 * 4;
    unsigned char *bgr = (unsigned char *)malloc(pitch * (size_t)h);
    if (bgr) {
        for (size_t i = 0; i < pitch * (size_t)h; i++) bgr[i] = (unsigned char)(i * 131u + 7u);
        float mean[3] = { 0.f, 0.f, 0.f }, norm[3] = { 1.f / 255.f, 1.f / 255.f, 1.f / 255.f };
        net_input(net, bgr, w, h, mean, norm);
        net_forward(net);                                  /* the stub has no device side: an error line, no crash */
        net_input(net, bgr, h > 1 ? h - 1 : 1, w > 8 ? 8 : w, mean, norm);     /* the other letterbox branch */
        free(bgr);
    }
    net_profile(net);
    net_free(net);
    return 0;
}
