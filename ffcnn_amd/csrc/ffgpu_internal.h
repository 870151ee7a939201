/* ffgpu_internal.h -- seam between the C host side (ffcnn_host.c) and the HIP
 * device side (ffgpu_*.hip) of libffcnn_hip.so.  Not installed. */
#ifndef FFGPU_INTERNAL_H
#define FFGPU_INTERNAL_H

#include "ffcnn.h"
#include "ffcnn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FFCNN_EXT_MAGIC 0x46464e58u   /* "FFNX" */

/* Private block net_load() places right behind the (layer_num + 1) LAYER
 * entries, inside the same allocation as NET (the reference allocates NET and
 * its LAYER array as one block too, ffcnn.c:123-126). */
typedef struct {
    unsigned magic;
    int      profile;          /* FFCNN_PROFILE=1: fill NET.timeused            */
    void    *dev;              /* ffgpu_netdev* (device weights + executors)    */
    BBOX    *own_boxes;        /* bbox_list storage (NOT aliased onto the input)*/
    int      box_cap;          /* boxes own_boxes has room for (= bbox_max at load) */
    float   *pinned_input;     /* layer_list[0].data when it is page-locked memory of the HIP runtime (ffgpu_host_alloc) */
} ffcnn_ext;

static inline ffcnn_ext *ffcnn_ext_of(NET *net)
{
    ffcnn_ext *e = (ffcnn_ext *)(net->layer_list + net->layer_num + 1);
    return e->magic == FFCNN_EXT_MAGIC ? e : (ffcnn_ext *)0;
}

/* implemented in ffgpu_exec.hip */
void *ffgpu_netdev_create(NET *net);              /* uploads weight_buf; NULL on failure */
void  ffgpu_netdev_destroy(void *dev);
int   ffgpu_netdev_forward1(NET *net, void *dev, int profile); /* one frame from layer_list[0].data -> bbox_list;  */
                                                  /* profile: per-kind device time added to net->timeused  */
int   ffgpu_netdev_profile_us(void *dev, double us_by_kind[LAYER_TYPE_TOTOAL]);   /* the same, in microseconds */
/* page-locked host memory owned by the HIP runtime (hipHostMalloc / hipHostFree), zero-filled; NULL on failure.  The input
 * tensor lives there: its upload is one DMA and no heap page is ever registered / unregistered */
float *ffgpu_host_alloc(size_t bytes);
void   ffgpu_host_free(float *p);
void  ffgpu_set_error(const char *fmt, ...);

#ifdef __cplusplus
}
#endif
#endif
