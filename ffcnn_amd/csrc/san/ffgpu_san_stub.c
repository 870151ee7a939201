/*
 * ffgpu_san_stub.c -- stand-in for the HIP side of ffgpu_internal.h in the SANITIZER build of the host code (make SAN=1;
 * tests/test_host_sanitizer.py).  NOT part of libffcnn_hip.so and no compute path: net_forward through it always fails.
 * It exists so that ffcnn_host.c -- the cfg / .weights parser, net_input, net_dump, net_free: the code that reads untrusted
 * text and bytes -- can run under -fsanitize=address,undefined on a machine without a GPU (the reference's
 * own latent UB: ffcnn.c:478-479, 261-264).
 *
 * FFSAN_NODEV=1: ffgpu_netdev_create fails (net_load's error path: net_free of a half-built net).
 * Otherwise a dummy handle is returned, so net_load hands the parsed NET to the caller; the stub walks every LAYER the way the
 * planner's first pass does (dependency indices, filter rows inside weight_buf) so out-of-range results of the parser are
 * touched -- and caught -- here.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ffgpu_internal.h"

static char g_err[512];

void ffgpu_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

const char *ffgpu_last_error(void) { return g_err; }

void *ffgpu_netdev_create(NET *net)
{
    if (getenv("FFSAN_NODEV")) { ffgpu_set_error("no HIP device visible (sanitizer stub, FFSAN_NODEV)"); return NULL; }
    /* what the device side reads of the host's result before any kernel exists */
    volatile float sink = 0.f;
    for (int i = 0; i < net->layer_num; i++) {
        const LAYER *l = net->layer_list + i;
        for (int k = 0; k < l->depend_num; k++) {
            const int d = l->depend_list[k];
            if (d < 0 || d >= net->layer_num) { ffgpu_set_error("layer %d depends on layer %d (of %d)", i, d, net->layer_num); return NULL; }
            sink += (float)net->layer_list[d + 1].c;
        }
        if (l->type == LAYER_TYPE_CONV && l->filter && l->fn > 0) {
            const int taps = l->fs * l->fs * (l->c / l->groups), rl = (taps + 3) / 4 * 4 + 4;
            sink += l->filter[0] + l->filter[(size_t)l->fn * rl - 1];          /* first and last float of the layer's rows */
        }
    }
    (void)sink;
    return malloc(16);
}

void ffgpu_netdev_destroy(void *dev) { free(dev); }

int ffgpu_netdev_forward1(NET *net, void *dev, int profile)
{
    (void)net; (void)dev; (void)profile;
    ffgpu_set_error("sanitizer stub: no device side");
    return -1;
}

int ffgpu_netdev_profile_us(void *dev, double us_by_kind[LAYER_TYPE_TOTOAL])
{
    (void)dev;
    for (int k = 0; k < LAYER_TYPE_TOTOAL; k++) us_by_kind[k] = 0.0;
    return 0;
}

float *ffgpu_host_alloc(size_t bytes) { return (float *)calloc(bytes ? bytes : 1, 1); }
void   ffgpu_host_free(float *p) { free(p); }
