/*
 * ffcnn_host.c -- host side (plain C) of libffcnn_hip.so: the ffcnn.h API.
 *
 *   net_load     darknet .cfg -> LAYER[] + shape inference, .weights -> folded
 *                filter rows, then hands the net to the device side
 *                (counterpart of reference ffcnn.c:114-247)
 *   net_input    BGR u8 -> planar fp32 letterbox (ffcnn.c:259-289)
 *   net_forward  one frame through the device executor (ffcnn.c:476-520)
 *   net_dump / net_profile / net_free
 *
 * All arithmetic of the forward path runs in HIP kernels (ffgpu_*.hip); this
 * file only parses, plans and moves bytes.  There is no CPU compute fallback:
 * net_load fails (NULL) when the device side cannot be created.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ffgpu_internal.h"

#define UP(x, n) (((x) + (n) - 1) / (n) * (n))

/* ------------------------------------------------------------------------ */
/* cfg text: the file is held in memory; a "section" is the span between one
 * '[' at line start and the next.  Keys are matched exactly at line start
 * (darknet semantics); the first occurrence in the section wins.           */

typedef struct { const char *name; size_t name_len; const char *body, *end; } cfg_sec;

static char *slurp(const char *path, size_t *len)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    char *buf = n >= 0 ? (char *)malloc((size_t)n + 2) : NULL;
    if (buf) {
        n = (long)fread(buf, 1, (size_t)n, fp);
        buf[n] = '\n'; buf[n + 1] = 0;
        if (len) *len = (size_t)n + 1;
    }
    fclose(fp);
    return buf;
}

static const char *skip_blank(const char *p) { while (*p == ' ' || *p == '\t' || *p == '\r') p++; return p; }
static const char *line_end(const char *p) { while (*p && *p != '\n') p++; return p; }

/* next section header at or after p; fills s, returns position after it or NULL */
static const char *next_section(const char *p, cfg_sec *s)
{
    while (*p) {
        const char *q = skip_blank(p);
        const char *e = line_end(q);
        if (*q == '[') {
            const char *close = memchr(q, ']', (size_t)(e - q));
            if (close) {
                s->name = q + 1; s->name_len = (size_t)(close - q - 1);
                s->body = *e ? e + 1 : e;
                const char *scan = s->body;       /* section ends at the next header line */
                for (;;) {
                    const char *l = skip_blank(scan);
                    if (!*l || *l == '[') { s->end = scan; break; }
                    scan = line_end(l); if (*scan) scan++;
                }
                return s->end;
            }
        }
        p = *e ? e + 1 : e;
    }
    return NULL;
}

static int sec_is(const cfg_sec *s, const char *name) { return strlen(name) == s->name_len && !memcmp(s->name, name, s->name_len); }

/* value of `key` in the section -> out (trimmed, NUL terminated); "" when absent */
static const char *sec_str(const cfg_sec *s, const char *key, char *out, size_t cap)
{
    size_t klen = strlen(key);
    out[0] = 0;
    for (const char *p = s->body; p < s->end; ) {
        const char *q = skip_blank(p), *e = line_end(q);
        if (*q != '#' && *q != ';' && (size_t)(e - q) > klen && !memcmp(q, key, klen)) {
            const char *v = skip_blank(q + klen);
            if (*v == '=') {
                v = skip_blank(v + 1);
                const char *t = e;
                while (t > v && (t[-1] == ' ' || t[-1] == '\t' || t[-1] == '\r')) t--;
                size_t n = (size_t)(t - v) < cap - 1 ? (size_t)(t - v) : cap - 1;
                memcpy(out, v, n); out[n] = 0;
                return out;
            }
        }
        p = *e ? e + 1 : e;
    }
    return out;
}

static int sec_num(const cfg_sec *s, const char *key) { char v[64]; return atoi(sec_str(s, key, v, sizeof v)); }

static int layer_kind(const cfg_sec *s)
{
    static const struct { const char *n; int k; } T[] = {
        { "convolutional", LAYER_TYPE_CONV }, { "conv", LAYER_TYPE_CONV },
        { "avgpool", LAYER_TYPE_AVGPOOL }, { "avg", LAYER_TYPE_AVGPOOL },
        { "maxpool", LAYER_TYPE_MAXPOOL }, { "max", LAYER_TYPE_MAXPOOL },
        { "upsample", LAYER_TYPE_UPSAMPLE }, { "dropout", LAYER_TYPE_DROPOUT },
        { "shortcut", LAYER_TYPE_SHORTCUT }, { "route", LAYER_TYPE_ROUTE }, { "yolo", LAYER_TYPE_YOLO } };
    for (size_t i = 0; i < sizeof T / sizeof T[0]; i++) if (sec_is(s, T[i].n)) return T[i].k;
    return -1;
}

static int activation_id(const char *v)
{
    if (!strncmp(v, "linear", 6)) return 0;
    if (!strncmp(v, "relu", 4)) return 1;
    if (!strncmp(v, "leaky", 5)) return 2;
    return -1;                                   /* treated as linear by every kernel */
}

static int split_ints(char *v, int *dst, int cap)
{
    int n = 0;
    for (char *t = strtok(v, ", \t"); t && n < cap; t = strtok(NULL, ", \t")) dst[n++] = atoi(t);
    return n;
}

/* ------------------------------------------------------------------------ */
static int filter_row_len(const LAYER *l) { return UP(l->fs * l->fs * (l->c / l->groups), 4) + 4; }

/* Limits on what a cfg may ask for.  The reference computes all of this in int without a check (ffcnn.c:139-171: a route or shortcut
 * index outside the net reads / writes beside layer_list, sizes overflow silently); here a cfg outside the limits makes net_load fail
 * with a message.  Every limit is far beyond any darknet cfg: tensors and the weight buffer must fit an int count of floats. */
#define FF_MAX_DIM   65536
#define FF_MAX_CH    (1 << 20)
#define FF_MAX_FLOATS 0x7fffffffLL

static int tensor_ok(const LAYER *t)
{
    return t->w >= 1 && t->h >= 1 && t->c >= 1 && t->w <= FF_MAX_DIM && t->h <= FF_MAX_DIM && t->c <= FF_MAX_CH &&
           (long long)t->w * t->h * t->c <= FF_MAX_FLOATS;
}

/* 0, or -1 with the error text set */
static int shape_layers(NET *net, const char *cfg, int inputw, int inputh)
{
    cfg_sec s;
    int cur = 0;
    char val[256];
    long long wsize = 0;
    if (inputw < 0 || inputh < 0 || inputw > FF_MAX_DIM || inputh > FF_MAX_DIM) { ffgpu_set_error("net_load: input size %d x %d", inputw, inputh); return -1; }
    for (const char *p = cfg; (p = next_section(p, &s)) != NULL; ) {
        if (sec_is(&s, "net")) {
            LAYER *l0 = net->layer_list;
            l0->w = inputw ? UP(inputw, 32) : sec_num(&s, "width");
            l0->h = inputh ? UP(inputh, 32) : sec_num(&s, "height");
            l0->c = sec_num(&s, "channels");
            continue;
        }
        int kind = layer_kind(&s);
        if (kind < 0 || cur >= net->layer_num) continue;
        LAYER *in = net->layer_list + cur, *out = in + 1;
        if (!tensor_ok(in)) {                                  /* (a head leaves its output zeroed: only a route may follow it) */
            if (!(kind == LAYER_TYPE_ROUTE && cur > 0 && in[-1].type == LAYER_TYPE_YOLO)) {
                ffgpu_set_error("net_load: layer %d has an input of %d x %d x %d", cur, in->w, in->h, in->c);
                return -1;
            }
        }
        in->type = kind; in->stride = 1; in->groups = 1;
        switch (kind) {
        case LAYER_TYPE_CONV: {
            int v;
            in->fn = sec_num(&s, "filters");
            in->fs = sec_num(&s, "size");
            if ((v = sec_num(&s, "stride")) != 0) in->stride = v;
            if ((v = sec_num(&s, "groups")) != 0) in->groups = v;
            in->pad = sec_num(&s, "pad") ? in->fs / 2 : 0;
            in->batchnorm = sec_num(&s, "batch_normalize") != 0;
            in->activation = activation_id(sec_str(&s, "activation", val, sizeof val));
            if (in->fn < 1 || in->fn > FF_MAX_CH || in->fs < 1 || in->fs > 1024 || in->stride < 1 || in->stride > FF_MAX_DIM || in->groups < 1 ||
                in->fs > in->w + 2 * in->pad || in->fs > in->h + 2 * in->pad || (long long)in->fs * in->fs * (in->c / in->groups) > FF_MAX_FLOATS / 8) {
                ffgpu_set_error("net_load: conv layer %d: filters %d size %d stride %d groups %d on %d x %d x %d", cur, in->fn, in->fs, in->stride, in->groups, in->w, in->h, in->c);
                return -1;
            }
            out->c = in->fn;
            out->w = (in->w + 2 * in->pad - in->fs) / in->stride + 1;
            out->h = (in->h + 2 * in->pad - in->fs) / in->stride + 1;
            wsize += (long long)in->fn * filter_row_len(in);
            if (wsize > FF_MAX_FLOATS) { ffgpu_set_error("net_load: more than 2^31 weights at layer %d", cur); return -1; }
            break; }
        case LAYER_TYPE_AVGPOOL: case LAYER_TYPE_MAXPOOL: {
            int v;
            in->fs = sec_num(&s, "size");
            if ((v = sec_num(&s, "stride")) != 0) in->stride = v;
            if (in->stride < 1 || in->fs < 0 || in->fs > FF_MAX_DIM) { ffgpu_set_error("net_load: pool layer %d: size %d stride %d", cur, in->fs, in->stride); return -1; }
            out->c = in->c; out->w = in->w / in->stride; out->h = in->h / in->stride;
            break; }
        case LAYER_TYPE_UPSAMPLE: {
            int v;
            if ((v = sec_num(&s, "stride")) != 0) in->stride = v;
            if (in->stride < 1 || (long long)in->w * in->stride > FF_MAX_DIM || (long long)in->h * in->stride > FF_MAX_DIM) { ffgpu_set_error("net_load: upsample layer %d: stride %d", cur, in->stride); return -1; }
            out->c = in->c; out->w = in->w * in->stride; out->h = in->h * in->stride;
            break; }
        case LAYER_TYPE_SHORTCUT: {
            const long long d = (long long)sec_num(&s, "from") + cur;
            if (d < 0 || d >= cur) { ffgpu_set_error("net_load: shortcut layer %d: from = layer %lld", cur, d); return -1; }
            in->depend_list[0] = (int)d;
            in->depend_num = 1;
            in->activation = activation_id(sec_str(&s, "activation", val, sizeof val));
            out->c = in->c; out->w = in->w; out->h = in->h;
            break; }
        case LAYER_TYPE_DROPOUT:
            out->c = in->c; out->w = in->w; out->h = in->h;
            break;
        case LAYER_TYPE_ROUTE: {
            int ids[4];
            int n = split_ints((char *)sec_str(&s, "layers", val, sizeof val), ids, 4);
            for (int k = 0; k < n; k++) {
                const long long d = ids[k] > 0 ? (long long)ids[k] : (long long)cur + ids[k];
                if (d < 0 || d >= cur) { ffgpu_set_error("net_load: route layer %d: source layer %lld", cur, d); return -1; }
                const LAYER *src = net->layer_list + d + 1;     /* OUTPUT of layer d */
                in->depend_list[k] = (int)d;
                if ((long long)out->c + src->c > FF_MAX_CH) { ffgpu_set_error("net_load: route layer %d: too many channels", cur); return -1; }
                out->c += src->c; out->w = src->w; out->h = src->h;
            }
            in->depend_num = n;
            break; }
        case LAYER_TYPE_YOLO: {
            int mask[9] = { 0 }, flat[18] = { 0 };
            in->class_num = sec_num(&s, "classes");
            sec_str(&s, "scale_x_y", val, sizeof val);
            in->scale_x_y = val[0] ? (float)atof(val) : 1.0f;
            in->ignore_thres = (float)atof(sec_str(&s, "ignore_thresh", val, sizeof val));
            split_ints((char *)sec_str(&s, "mask", val, sizeof val), mask, 9);
            split_ints((char *)sec_str(&s, "anchors", val, sizeof val), flat, 18);
            for (int k = 0; k < 3; k++) {
                int m = mask[k] >= 0 && mask[k] < 9 ? mask[k] : 0;
                in->anchor_list[k][0] = flat[2 * m];
                in->anchor_list[k][1] = flat[2 * m + 1];
            }
            break; }                               /* a head leaves layer_list[cur+1] zeroed */
        }
        cur++;
    }
    if (!tensor_ok(net->layer_list)) { ffgpu_set_error("net_load: empty or oversized input geometry %d x %d x %d", net->layer_list->w, net->layer_list->h, net->layer_list->c); return -1; }
    net->weight_size = (int)wsize;
    return 0;
}

static int count_layers(const char *cfg)
{
    cfg_sec s; int n = 0;
    for (const char *p = cfg; (p = next_section(p, &s)) != NULL; ) if (layer_kind(&s) >= 0) n++;
    return n;
}

/* darknet .weights: 20-byte header {int32 major, minor, revision; uint64 seen},
 * then per conv layer: fn biases | if BN: fn scales, fn means, fn variances |
 * fn*K taps ([fn][c/groups][fs][fs]).  Rows are stored padded:
 * [taps.., 0.., scale', bias', mean, var] with the batch-norm folded in. */
static void read_weights(NET *net, const char *path)
{
    FILE *fp = path ? fopen(path, "rb") : NULL;
    float *row0 = net->weight_buf;
    int short_warned = 0;
    if (fp) fseek(fp, 20, SEEK_SET);
    for (int i = 0; i < net->layer_num; i++) {
        LAYER *l = net->layer_list + i;
        if (l->type != LAYER_TYPE_CONV) continue;
        const int rl = filter_row_len(l), taps = l->fs * l->fs * (l->c / l->groups);
        float *tail = row0 + rl - 4;              /* -> scale' of row 0 */
        l->filter = row0;
        row0 += (size_t)l->fn * rl;
        if (!fp) continue;
        size_t got = 0, want = (size_t)l->fn * (size_t)(1 + (l->batchnorm ? 3 : 0) + taps);
        for (int j = 0; j < l->fn; j++) { tail[j * rl] = 1.0f; got += fread(&tail[j * rl + 1], sizeof(float), 1, fp); }
        if (l->batchnorm) {
            static const int slot[3] = { 0, 2, 3 };             /* scale, mean, variance */
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < l->fn; j++) got += fread(&tail[j * rl + slot[k]], sizeof(float), 1, fp);
            for (int j = 0; j < l->fn; j++) {
                float *t = tail + j * rl;
                t[0] /= (float)sqrt(t[3] + 0.00001f);
                t[1] -= t[2] * t[0];
            }
        }
        for (int j = 0; j < l->fn; j++) got += fread(l->filter + (size_t)j * rl, sizeof(float), (size_t)taps, fp);
        /* the reference ignores short reads too (ffcnn.c:211-235) and runs on with zero filters; say so at least */
        if (got != want && !short_warned) {
            short_warned = 1;
            fprintf(stderr, "ffcnn: weights file '%s' ends inside layer %d (%zu of %zu floats): remaining filters stay zero\n",
                    path, i, got, want);
        }
    }
    if (fp) fclose(fp);
}

NET *net_load(char *cfg_path, char *weights_path, int inputw, int inputh)
{
    char *cfg = cfg_path ? slurp(cfg_path, NULL) : NULL;
    if (!cfg) { ffgpu_set_error("net_load: cannot read cfg '%s'", cfg_path ? cfg_path : "(null)"); return NULL; }
    const int nl = count_layers(cfg);
    NET *net = (NET *)calloc(1, sizeof(NET) + (size_t)(nl + 1) * sizeof(LAYER) + sizeof(ffcnn_ext));
    if (!net) { free(cfg); return NULL; }
    net->layer_list = (LAYER *)(net + 1);
    net->layer_num = nl;
    const int shaped = shape_layers(net, cfg, inputw, inputh);
    free(cfg);
    if (shaped != 0) { free(net); return NULL; }               /* (nothing but the block itself exists yet) */

    ffcnn_ext *ext = (ffcnn_ext *)(net->layer_list + nl + 1);
    ext->magic = FFCNN_EXT_MAGIC;
    const char *prof = getenv("FFCNN_PROFILE");
    ext->profile = prof && atoi(prof) != 0;

    LAYER *l0 = net->layer_list;
    size_t in_floats = (size_t)l0->w * l0->h * l0->c;
    net->weight_buf = (float *)calloc(net->weight_size > 0 ? (size_t)net->weight_size : 1, sizeof(float));
    l0->data = (float *)calloc(in_floats ? in_floats : 1, sizeof(float));
    net->bbox_max = (int)(in_floats * sizeof(float) / sizeof(BBOX));       /* same capacity as ffcnn.c:243 (51 200 at 320x320) */
    ext->own_boxes = (BBOX *)calloc((size_t)(net->bbox_max > 0 ? net->bbox_max : 1), sizeof(BBOX));
    ext->box_cap = net->bbox_max > 0 ? net->bbox_max : 1;
    net->bbox_list = ext->own_boxes;
    if (!net->weight_buf || !l0->data || !ext->own_boxes || in_floats == 0) {
        ffgpu_set_error("net_load: allocation failed or empty input geometry");
        net_free(net);
        return NULL;
    }
    read_weights(net, weights_path);

    ext->dev = ffgpu_netdev_create(net);
    if (!ext->dev) { net_free(net); return NULL; }          /* no device, no net: error text already set */
    /* the tensor the application fills (net_input, or its own writes) moves into page-locked memory of the HIP runtime: net_forward
     * uploads it with one DMA.  Best effort -- a failure leaves the calloc'd tensor in place and only costs time. */
    ext->pinned_input = ffgpu_host_alloc(in_floats * sizeof(float));
    if (ext->pinned_input) { free(l0->data); l0->data = ext->pinned_input; }
    return net;
}

void net_free(NET *net)
{
    if (!net) return;
    ffcnn_ext *ext = ffcnn_ext_of(net);
    if (ext) {
        if (ext->dev) ffgpu_netdev_destroy(ext->dev);
        if (ext->pinned_input) {                             /* goes back to the runtime, not to free() */
            if (net->layer_list[0].data == ext->pinned_input) net->layer_list[0].data = NULL;
            ffgpu_host_free(ext->pinned_input);
            ext->pinned_input = NULL;
        }
        free(ext->own_boxes);
        ext->magic = 0;
    }
    for (int i = 0; i <= net->layer_num; i++) free(net->layer_list[i].data);
    free(net->cnntempbuf);
    free(net->weight_buf);
    free(net);
}

void net_input(NET *net, unsigned char *bgr, int w, int h, float *mean, float *norm)
{
    if (!net || !bgr || w <= 0 || h <= 0) return;
    LAYER *l0 = net->layer_list;
    const int W = l0->w, H = l0->h;
    int sw, sh;
    memset(net->bbox_list, 0, sizeof(BBOX) * (size_t)net->bbox_num);
    net->bbox_num = 0;
    if ((long)w * H > (long)h * W) { sw = W; sh = (int)((long)sw * h / w); net->s1 = w; net->s2 = sw; }
    else                           { sh = H; sw = (int)((long)sh * w / h); net->s1 = h; net->s2 = sh; }
    const size_t pitch = (size_t)UP(w * 3, 4), plane = (size_t)W * H;
    float *dst = l0->data;
    for (int y = 0; y < sh; y++) {
        const unsigned char *src = bgr + (size_t)((long)y * net->s1 / net->s2) * pitch;
        float *o = dst + (size_t)y * W;
        for (int x = 0; x < sw; x++) {
            const unsigned char *px = src + (size_t)((long)x * net->s1 / net->s2) * 3;
            o[x]             = (px[2] - mean[0]) * norm[0];
            o[x + plane]     = (px[1] - mean[1]) * norm[1];
            o[x + 2 * plane] = (px[0] - mean[2]) * norm[2];
        }
    }
}

void net_forward(NET *net)
{
    if (!net) return;
    ffcnn_ext *ext = ffcnn_ext_of(net);
    if (!ext || !ext->dev) { fprintf(stderr, "ffcnn: net_forward on a net without device state\n"); return; }
    if (ffgpu_netdev_forward1(net, ext->dev, ext->profile) != 0)
        fprintf(stderr, "ffcnn: net_forward failed: %s\n", ffgpu_last_error());
}

static const char *kind_name(int k)
{
    static const char *N[] = { "conv", "avgpool", "maxpool", "upsample", "dropout", "shortcut", "route", "yolo" };
    return (k >= 0 && k < LAYER_TYPE_TOTOAL) ? N[k] : "unknown";
}

static const char *act_name(int a)
{
    return a == 0 ? "linear" : a == 1 ? "relu" : a == 2 ? "leaky" : "unknown";
}

void net_dump(NET *net)
{
    if (!net) return;
    printf("layer   type  filters fltsize  pad/strd input          output       bn/act\n");
    for (int i = 0; i < net->layer_num; i++) {
        const LAYER *a = net->layer_list + i, *b = a + 1;
        switch (a->type) {
        case LAYER_TYPE_YOLO:
            printf("%3d %8s class_num: %d ignore_thres: %3.2f [%d, %d] [%d, %d] [%d, %d]\n", i, kind_name(a->type),
                   a->class_num, a->ignore_thres, a->anchor_list[0][0], a->anchor_list[0][1],
                   a->anchor_list[1][0], a->anchor_list[1][1], a->anchor_list[2][0], a->anchor_list[2][1]);
            break;
        case LAYER_TYPE_DROPOUT: case LAYER_TYPE_SHORTCUT: case LAYER_TYPE_ROUTE: {
            char deps[256] = "";
            if (a->type != LAYER_TYPE_DROPOUT) {
                size_t o = (size_t)snprintf(deps, sizeof deps, "layers:");
                for (int k = 0; k < a->depend_num && o < sizeof deps; k++)
                    o += (size_t)snprintf(deps + o, sizeof deps - o, " %d", a->depend_list[k]);
            }
            printf("%3d %8s %-38s -> %3dx%3dx%3d\n", i, kind_name(a->type), deps, b->w, b->h, b->c);
            break; }
        default:
            printf("%3d %8s %3d/%3d %2dx%2dx%3d   %d/%2d   %3dx%3dx%3d -> %3dx%3dx%3d  %d/%-6s\n", i, kind_name(a->type),
                   a->fn, a->groups, a->fs, a->fs, a->c / a->groups, a->pad, a->stride, a->w, a->h, a->c,
                   b->w, b->h, b->c, a->batchnorm, act_name(a->activation));
        }
    }
}

void net_profile(NET *net)
{
    if (!net) return;
    /* same lines as ffcnn.c:550; timeused[] is filled under FFCNN_PROFILE=1 (the reference: ENABLE_NET_PROFILE) from
     * device time -- whole milliseconds of the time accumulated over all net_forward calls so far */
    for (int k = 0; k < LAYER_TYPE_TOTOAL; k++) printf("%8s: %5d ms\n", kind_name(k), net->timeused[k]);
    ffcnn_ext *ext = ffcnn_ext_of(net);
    double us[LAYER_TYPE_TOTOAL];
    if (ext && ext->profile && ext->dev && getenv("FFCNN_PROFILE_US") && ffgpu_netdev_profile_us(ext->dev, us) == 0)
        for (int k = 0; k < LAYER_TYPE_TOTOAL; k++) printf("%8s: %9.1f us\n", kind_name(k), us[k]);
}
