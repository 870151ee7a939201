/*
 * ffcnn_oracle.c -- CPU restatement of the ffcnn forward path (TEST INFRASTRUCTURE,
 * see ffcnn_oracle.h).  Build with -O2 -ffp-contract=off (oracle/Makefile): with
 * those flags it is bit-identical to the reference's conv-v0..v5 family built the
 * same way (checked by tests/test_oracle_vs_ref.py).
 *
 * Pinned against: oracle/_ref (unmodified reference compiled from /root/reference)
 * and tests/golden/.  All file:line citations are into /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ffcnn_oracle.h"

static int round_up(int x, int n) { return (x + n - 1) / n * n; }

/* utils.h:15-23 -- linear / relu / leaky 0.1 / sigmoid via double exp */
float orc_activate(float x, int act)
{
    if (act == 1) return x > 0 ? x : 0;
    if (act == 2) return x > 0 ? x : 0.1f * x;
    if (act == 3) return 1.0f / (1.0f + (float)exp(-x));
    return x;
}

/*
 * conv-v0.c:7-31 (one group) + conv-v0.c:36-52 (group loop).  Accumulation
 * order per output: input channel, then tap row, then tap column, starting
 * from 0 and skipping out-of-image taps -- the order every v0..v5 variant
 * reproduces.  Epilogue: act(sum * row[K4] + row[K4+1]).
 */
void orc_groupconv(const float *in, const float *filt, float *out,
                   int iw, int ih, int ic, int groups, int pad, int stride,
                   int fs, int fn, int ow, int oh, int oc, int act, int compat_v6)
{
    const int gic = ic / groups, goc = oc / groups;
    const int k4 = round_up(fs * fs * gic, 4), row = k4 + 4;
    /* conv-v6.c:499-502 dispatch condition for the defective 5x5 path */
    const int v6_dw5 = compat_v6 && pad == 2 && fs == 5 && stride == 1 && gic == 1;
    (void)fn;
    for (int o = 0; o < oc; o++) {
        const int g = o / goc;
        const float *w = filt + (size_t)o * row;
        const float *src = in + (size_t)g * gic * iw * ih;
        float *dst = out + (size_t)o * ow * oh;
        for (int y = 0; y < oh; y++) {
            for (int x = 0; x < ow; x++) {
                float acc = 0;
                for (int ci = 0; ci < gic; ci++) {
                    for (int ky = 0; ky < fs; ky++) {
                        /* conv-v6.c:422-441: output row oh-2 never reads tap row 0 */
                        if (v6_dw5 && oh > 2 && y == oh - 2 && ky == 0) continue;
                        const int sy = y * stride - pad + ky;
                        if (sy < 0 || sy >= ih) continue;
                        for (int kx = 0; kx < fs; kx++) {
                            const int sx = x * stride - pad + kx;
                            if (sx < 0 || sx >= iw) continue;
                            acc += src[(size_t)ci * iw * ih + (size_t)sy * iw + sx] * w[ci * fs * fs + ky * fs + kx];
                        }
                    }
                }
                dst[y * ow + x] = orc_activate(acc * w[k4] + w[k4 + 1], act);
            }
        }
    }
}

/* ffcnn.c:337-372 (window [x-(fs-1)/2, +fs) clipped to the plane) and
 * ffcnn.c:381-394 (outputs sampled at 0, stride, 2*stride, ...). */
void orc_pool(const float *in, float *out, int w, int h, int c, int fs, int stride, int is_max)
{
    const int ow = w / stride, oh = h / stride;
    for (int ch = 0; ch < c; ch++) {
        const float *p = in + (size_t)ch * w * h;
        for (int oy = 0; oy < oh; oy++) for (int ox = 0; ox < ow; ox++) {
            int x0 = ox * stride - (fs - 1) / 2, y0 = oy * stride - (fs - 1) / 2;
            int x1 = x0 + fs, y1 = y0 + fs;
            if (x0 < 0) x0 = 0;
            if (y0 < 0) y0 = 0;
            if (x1 > w) x1 = w;
            if (y1 > h) y1 = h;
            float v = is_max ? p[y0 * w + x0] : 0;
            for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) {
                if (is_max) { if (v < p[y * w + x]) v = p[y * w + x]; }
                else v += p[y * w + x];
            }
            out[((size_t)ch * oh + oy) * ow + ox] = is_max ? v : v / (fs * fs);
        }
    }
}

/* ffcnn.c:396-410 -- nearest-neighbour replicate by `stride` */
void orc_upsample(const float *in, float *out, int w, int h, int c, int stride)
{
    const int ow = w * stride, oh = h * stride;
    for (int ch = 0; ch < c; ch++)
        for (int y = 0; y < oh; y++)
            for (int x = 0; x < ow; x++)
                out[((size_t)ch * oh + y) * ow + x] = in[((size_t)ch * h + y / stride) * w + x / stride];
}

/* ffcnn.c:418-423 */
void orc_shortcut(const float *a, const float *b, float *out, int n, int act)
{
    for (int i = 0; i < n; i++) out[i] = orc_activate(a[i] + b[i], act);
}

/*
 * ffcnn.c:438-474.  Per cell, per anchor k: objectness logit bs at channel
 * k*(5+classes)+4, best class = first maximum of the raw class logits, and
 * conf = 1 / (1 + exp(-bs) * (1 + exp(-cs)))  exactly as parenthesised at
 * ffcnn.c:451 (exp in double, narrowed to float before the multiply).
 */
void orc_yolo(const float *in, int w, int h, int classes, const int anchors[3][2],
              float thresh, float scale_xy, int netw, int neth,
              orc_box *cand, int *ncand, int cap)
{
    const size_t hw = (size_t)w * h;
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) for (int k = 0; k < 3; k++) {
        const float *cell = in + (size_t)k * (5 + classes) * hw + (size_t)i * w + j;
        float bs = cell[4 * hw], cs = cell[5 * hw];
        int best = 0;
        for (int l = 1; l < classes; l++) {
            float v = cell[(5 + l) * hw];
            if (cs < v) { cs = v; best = l; }
        }
        float conf = 1.0f / ((1.0f + (float)exp(-bs) * (1.0f + (float)exp(-cs))));
        if (!(conf >= thresh)) continue;
        float tx = cell[0], ty = cell[hw], tw = cell[2 * hw], th = cell[3 * hw];
        float cx = (j + orc_activate(tx, 3)) * netw / w;
        float cy = (i + orc_activate(ty, 3)) * neth / h;
        float bw = (float)exp(tw) * anchors[k][0] * scale_xy;
        float bh = (float)exp(th) * anchors[k][1] * scale_xy;
        if (*ncand < cap) {
            orc_box *b = cand + (*ncand)++;
            b->type = best; b->score = conf;
            b->x1 = cx - bw * 0.5f; b->y1 = cy - bh * 0.5f;
            b->x2 = cx + bw * 0.5f; b->y2 = cy + bh * 0.5f;
        }
    }
}

static int by_score_desc(const void *a, const void *b)
{
    float sa = ((const orc_box *)a)->score, sb = ((const orc_box *)b)->score;
    return sa < sb ? 1 : sa > sb ? -1 : 0;
}

/*
 * ffcnn.c:298-335.  qsort by score (same libc, same comparator => same order on
 * ties), greedy suppression among equal classes with
 * metric = inter / min(area) when use_min else inter / union, suppress if > thresh;
 * survivors compacted and scaled coord * s1 / s2; tail zeroed.
 */
int orc_nms(orc_box *b, int n, float thresh, int use_min, int s1, int s2)
{
    if (!b || n <= 0) return 0;
    qsort(b, n, sizeof(*b), by_score_desc);
    for (int c = 0; c < n; c++) {
        if (b[c].score == 0) continue;           /* suppressed boxes never suppress */
        for (int j = c + 1; j < n; j++) {
            if (b[j].score == 0 || b[j].type != b[c].type) continue;
            float xa = b[c].x1 > b[j].x1 ? b[c].x1 : b[j].x1;
            float ya = b[c].y1 > b[j].y1 ? b[c].y1 : b[j].y1;
            float xb = b[c].x2 < b[j].x2 ? b[c].x2 : b[j].x2;
            float yb = b[c].y2 < b[j].y2 ? b[c].y2 : b[j].y2;
            float inter = (xa < xb && ya < yb) ? (xb - xa) * (yb - ya) : 0;
            float ac = (b[c].x2 - b[c].x1) * (b[c].y2 - b[c].y1);
            float aj = (b[j].x2 - b[j].x1) * (b[j].y2 - b[j].y1);
            float uni = ac + aj - inter;
            float m = use_min ? inter / (ac < aj ? ac : aj) : inter / uni;
            if (m > thresh) b[j].score = 0;
        }
    }
    int keep = 0;
    for (int i = 0; i < n; i++) {
        if (!b[i].score) continue;
        orc_box t = b[i];
        b[keep].type = t.type; b[keep].score = t.score;
        b[keep].x1 = t.x1 * s1 / s2; b[keep].y1 = t.y1 * s1 / s2;
        b[keep].x2 = t.x2 * s1 / s2; b[keep].y2 = t.y2 * s1 / s2;
        keep++;
    }
    memset(b + keep, 0, sizeof(*b) * (size_t)(n - keep));
    return keep;
}

/* ----------------------------------------------------------------------- */
/* cfg: line-based section/key=value reader (semantics of ffcnn.c:128-208)  */

typedef struct { char key[40]; char val[256]; } kv_t;
typedef struct { char name[32]; int n; kv_t kv[48]; } section_t;

static char *trim(char *s)
{
    while (*s == ' ' || *s == '\t' || *s == '\r') s++;
    char *e = s + strlen(s);
    while (e > s && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r' || e[-1] == '\n')) *--e = 0;
    return s;
}

static const char *sec_get(const section_t *s, const char *key)
{
    for (int i = 0; i < s->n; i++) if (!strcmp(s->kv[i].key, key)) return s->kv[i].val;
    return "";
}

static int sec_int(const section_t *s, const char *key) { return atoi(sec_get(s, key)); }

static int act_code(const char *s)       /* ffcnn.c:86-93: prefix match, else -1 */
{
    if (!strncmp(s, "linear", 6)) return 0;
    if (!strncmp(s, "relu", 4)) return 1;
    if (!strncmp(s, "leaky", 5)) return 2;
    return -1;
}

static int kind_of(const char *name)     /* ffcnn.c:52 accepted section names */
{
    if (!strcmp(name, "conv") || !strcmp(name, "convolutional")) return ORC_CONV;
    if (!strcmp(name, "avg") || !strcmp(name, "avgpool")) return ORC_AVGPOOL;
    if (!strcmp(name, "max") || !strcmp(name, "maxpool")) return ORC_MAXPOOL;
    if (!strcmp(name, "upsample")) return ORC_UPSAMPLE;
    if (!strcmp(name, "dropout")) return ORC_DROPOUT;
    if (!strcmp(name, "shortcut")) return ORC_SHORTCUT;
    if (!strcmp(name, "route")) return ORC_ROUTE;
    if (!strcmp(name, "yolo")) return ORC_YOLO;
    return -1;
}

static int read_sections(const char *path, section_t **out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    int cap = 64, n = 0;
    section_t *secs = calloc(cap, sizeof(*secs));
    char line[1024];
    while (fgets(line, sizeof line, fp)) {
        char *s = trim(line);
        if (!*s || *s == '#' || *s == ';') continue;
        if (*s == '[') {
            char *e = strchr(s, ']');
            if (!e) continue;
            *e = 0;
            if (n == cap) { cap *= 2; secs = realloc(secs, cap * sizeof(*secs)); }
            memset(&secs[n], 0, sizeof(secs[n]));
            snprintf(secs[n].name, sizeof secs[n].name, "%s", s + 1);
            n++;
        } else if (n > 0) {
            char *eq = strchr(s, '=');
            section_t *sec = &secs[n - 1];
            if (!eq || sec->n >= 48) continue;
            *eq = 0;
            snprintf(sec->kv[sec->n].key, sizeof sec->kv[0].key, "%s", trim(s));
            snprintf(sec->kv[sec->n].val, sizeof sec->kv[0].val, "%s", trim(eq + 1));
            sec->n++;
        }
    }
    fclose(fp);
    *out = secs;
    return n;
}

orc_net *orc_load(const char *cfg, const char *weights, int inputw, int inputh)
{
    section_t *secs = NULL;
    int nsec = read_sections(cfg, &secs);
    if (nsec < 0) return NULL;
    orc_net *net = calloc(1, sizeof(*net));
    int nl = 0;
    for (int i = 0; i < nsec; i++) if (kind_of(secs[i].name) >= 0) nl++;
    net->layers = calloc(nl ? nl : 1, sizeof(orc_layer));
    net->nlayers = nl;

    int cw = 0, ch = 0, cc = 0, li = 0;           /* running tensor geometry */
    for (int i = 0; i < nsec; i++) {
        const section_t *s = &secs[i];
        if (!strcmp(s->name, "net")) {            /* ffcnn.c:132-136 */
            cw = inputw ? round_up(inputw, 32) : sec_int(s, "width");
            ch = inputh ? round_up(inputh, 32) : sec_int(s, "height");
            cc = sec_int(s, "channels");
            net->in_w = cw; net->in_h = ch; net->in_c = cc;
            continue;
        }
        int kind = kind_of(s->name);
        if (kind < 0) continue;
        orc_layer *L = &net->layers[li];
        L->kind = kind; L->iw = cw; L->ih = ch; L->ic = cc; L->stride = 1; L->groups = 1;
        switch (kind) {
        case ORC_CONV:                            /* ffcnn.c:137-150 */
            L->fn = sec_int(s, "filters"); L->fs = sec_int(s, "size");
            if (sec_int(s, "stride")) L->stride = sec_int(s, "stride");
            if (sec_int(s, "groups")) L->groups = sec_int(s, "groups");
            L->pad = sec_int(s, "pad") ? L->fs / 2 : 0;
            L->batchnorm = !!sec_int(s, "batch_normalize");
            L->act = act_code(sec_get(s, "activation"));
            L->oc = L->fn;
            L->ow = (cw - L->fs + 2 * L->pad) / L->stride + 1;
            L->oh = (ch - L->fs + 2 * L->pad) / L->stride + 1;
            net->nweights += L->fn * (round_up(L->fs * L->fs * (cc / L->groups), 4) + 4);
            break;
        case ORC_AVGPOOL: case ORC_MAXPOOL:       /* ffcnn.c:151-157 */
            L->fs = sec_int(s, "size");
            if (sec_int(s, "stride")) L->stride = sec_int(s, "stride");
            L->oc = cc; L->ow = cw / L->stride; L->oh = ch / L->stride;
            break;
        case ORC_UPSAMPLE:                        /* ffcnn.c:158-163 */
            if (sec_int(s, "stride")) L->stride = sec_int(s, "stride");
            L->oc = cc; L->ow = cw * L->stride; L->oh = ch * L->stride;
            break;
        case ORC_SHORTCUT:                        /* ffcnn.c:168-171 */
            L->dep[0] = sec_int(s, "from") + li; L->ndep = 1;
            L->act = act_code(sec_get(s, "activation"));
            /* fallthrough */
        case ORC_DROPOUT:
            L->oc = cc; L->ow = cw; L->oh = ch;
            break;
        case ORC_ROUTE: {                         /* ffcnn.c:174-186 */
            char tmp[256]; snprintf(tmp, sizeof tmp, "%s", sec_get(s, "layers"));
            int k = 0;
            for (char *t = strtok(tmp, ","); t && k < 4; t = strtok(NULL, ","), k++) {
                int d = atoi(t);
                d = d > 0 ? d : li + d;
                L->dep[k] = d;
                L->oc += net->layers[d].oc; L->ow = net->layers[d].ow; L->oh = net->layers[d].oh;
            }
            L->ndep = k;
            break; }
        case ORC_YOLO: {                          /* ffcnn.c:187-204 */
            int mask[9] = {0}, anc[9][2] = {{0}};
            char tmp[256];
            L->classes = sec_int(s, "classes");
            L->scale_xy = *sec_get(s, "scale_x_y") ? (float)atof(sec_get(s, "scale_x_y")) : 1.0f;
            L->thresh = (float)atof(sec_get(s, "ignore_thresh"));
            snprintf(tmp, sizeof tmp, "%s", sec_get(s, "mask"));
            int k = 0;
            for (char *t = strtok(tmp, ","); t && k < 9; t = strtok(NULL, ","), k++) mask[k] = atoi(t);
            snprintf(tmp, sizeof tmp, "%s", sec_get(s, "anchors"));
            k = 0;
            for (char *t = strtok(tmp, ","); t && k < 18; t = strtok(NULL, ","), k++) anc[k / 2][k % 2] = atoi(t);
            for (k = 0; k < 3; k++) { L->anchors[k][0] = anc[mask[k]][0]; L->anchors[k][1] = anc[mask[k]][1]; }
            L->oc = L->ow = L->oh = 0;            /* a head has no output tensor (layer_list[i+1] stays zeroed) */
            break; }
        }
        cw = L->ow; ch = L->oh; cc = L->oc;
        li++;
    }
    free(secs);

    /* weights: ffcnn.c:107-112 (20-byte header) and 211-239 (per conv layer:
     * fn biases | [fn scales, fn means, fn variances] | fn*K taps; BN folded as
     * scale' = scale / sqrt(var + 1e-5f), bias' = bias - mean * scale'). */
    net->weights = calloc(net->nweights ? net->nweights : 1, sizeof(float));
    FILE *fp = weights ? fopen(weights, "rb") : NULL;
    if (fp) fseek(fp, 20, SEEK_SET);
    float *wp = net->weights;
    for (int i = 0; i < nl; i++) {
        orc_layer *L = &net->layers[i];
        if (L->kind != ORC_CONV) continue;
        const int K = L->fs * L->fs * (L->ic / L->groups), row = round_up(K, 4) + 4;
        L->filt = wp; wp += (size_t)L->fn * row;
        if (!fp) continue;
        size_t got = 0;
        for (int j = 0; j < L->fn; j++) {
            L->filt[j * row + row - 4] = 1.0f;
            got += fread(&L->filt[j * row + row - 3], 4, 1, fp);
        }
        if (L->batchnorm) {
            for (int j = 0; j < L->fn; j++) got += fread(&L->filt[j * row + row - 4], 4, 1, fp);
            for (int j = 0; j < L->fn; j++) got += fread(&L->filt[j * row + row - 2], 4, 1, fp);
            for (int j = 0; j < L->fn; j++) got += fread(&L->filt[j * row + row - 1], 4, 1, fp);
            for (int j = 0; j < L->fn; j++) {
                float *r = &L->filt[j * row + row - 4];
                r[0] /= (float)sqrt(r[3] + 0.00001f);
                r[1] -= r[2] * r[0];
            }
        }
        for (int j = 0; j < L->fn; j++) got += fread(&L->filt[j * row], 4, K, fp);
        net->weights_consumed += (int)got;
    }
    if (fp) fclose(fp);

    net->input = calloc((size_t)net->in_w * net->in_h * net->in_c, sizeof(float));
    for (int i = 0; i < nl; i++) {
        orc_layer *L = &net->layers[i];
        if (L->kind == ORC_YOLO || L->kind == ORC_DROPOUT) continue;
        L->out = calloc((size_t)L->ow * L->oh * L->oc, sizeof(float));
    }
    net->cap = (int)((size_t)net->in_w * net->in_h * net->in_c * sizeof(float) / sizeof(orc_box)); /* ffcnn.c:243 */
    net->cand = calloc(net->cap ? net->cap : 1, sizeof(orc_box));
    net->boxes = calloc(net->cap ? net->cap : 1, sizeof(orc_box));
    return net;
}

void orc_free(orc_net *n)
{
    if (!n) return;
    for (int i = 0; i < n->nlayers; i++) free(n->layers[i].out);
    free(n->layers); free(n->weights); free(n->input); free(n->cand); free(n->boxes); free(n);
}

/* ffcnn.c:259-289 */
void orc_input(orc_net *n, const unsigned char *bgr, int w, int h, const float mean[3], const float norm[3])
{
    const int W = n->in_w, H = n->in_h;
    int sw, sh;
    n->nboxes = 0;
    if (w * H > h * W) { sw = W; sh = sw * h / w; n->s1 = w; n->s2 = sw; }
    else               { sh = H; sw = sh * w / h; n->s1 = h; n->s2 = sh; }
    const int pitch = round_up(w * 3, 4);
    float *r = n->input, *g = r + (size_t)W * H, *b = g + (size_t)W * H;
    for (int i = 0; i < sh; i++) for (int j = 0; j < sw; j++) {
        const int x = j * n->s1 / n->s2, y = i * n->s1 / n->s2;
        const unsigned char *px = bgr + (size_t)y * pitch + x * 3;
        r[i * W + j] = (px[2] - mean[0]) * norm[0];
        g[i * W + j] = (px[1] - mean[1]) * norm[1];
        b[i * W + j] = (px[0] - mean[2]) * norm[2];
    }
}

const float *orc_layer_out(const orc_net *n, int i)
{
    if (i < 0) return n->input;
    const orc_layer *L = &n->layers[i];
    if (L->kind == ORC_DROPOUT) return orc_layer_out(n, i - 1);   /* ffcnn.c:412-416: pointer move */
    return L->out;
}

/* ffcnn.c:476-520 with every activation retained */
void orc_forward(orc_net *n, int compat_v6)
{
    n->ncand = 0;
    for (int i = 0; i < n->nlayers; i++) {
        orc_layer *L = &n->layers[i];
        const float *in = orc_layer_out(n, i - 1);
        switch (L->kind) {
        case ORC_CONV:
            orc_groupconv(in, L->filt, L->out, L->iw, L->ih, L->ic, L->groups, L->pad, L->stride,
                          L->fs, L->fn, L->ow, L->oh, L->oc, L->act, compat_v6);
            break;
        case ORC_AVGPOOL: orc_pool(in, L->out, L->iw, L->ih, L->ic, L->fs, L->stride, 0); break;
        case ORC_MAXPOOL: orc_pool(in, L->out, L->iw, L->ih, L->ic, L->fs, L->stride, 1); break;
        case ORC_UPSAMPLE: orc_upsample(in, L->out, L->iw, L->ih, L->ic, L->stride); break;
        case ORC_SHORTCUT:
            orc_shortcut(in, orc_layer_out(n, L->dep[0]), L->out, L->ow * L->oh * L->oc, L->act);
            break;
        case ORC_ROUTE: {                         /* ffcnn.c:425-434 */
            float *dst = L->out;
            for (int k = 0; k < L->ndep; k++) {
                const orc_layer *S = &n->layers[L->dep[k]];
                size_t cnt = (size_t)S->ow * S->oh * S->oc;
                memcpy(dst, orc_layer_out(n, L->dep[k]), cnt * sizeof(float));
                dst += cnt;
            }
            break; }
        case ORC_YOLO:
            orc_yolo(in, L->iw, L->ih, L->classes, (const int (*)[2])L->anchors, L->thresh, L->scale_xy,
                     n->in_w, n->in_h, n->cand, &n->ncand, n->cap);
            break;
        default: break;
        }
    }
    memcpy(n->boxes, n->cand, sizeof(orc_box) * (size_t)n->ncand);
    n->nboxes = orc_nms(n->boxes, n->ncand, 0.5f, 1, n->s1, n->s2);   /* ffcnn.c:519 */
}

/* text of ffcnn.c:522-548 */
int orc_dump(const orc_net *n, char *buf, int len)
{
    static const char *KIND[] = { "conv", "avgpool", "maxpool", "upsample", "dropout", "shortcut", "route", "yolo" };
    static const char *ACT[] = { "linear", "relu", "leaky" };
    int o = 0;
#define EMIT(...) do { int r_ = snprintf(buf + o, o < len ? (size_t)(len - o) : 0, __VA_ARGS__); if (r_ > 0) o += r_; } while (0)
    EMIT("layer   type  filters fltsize  pad/strd input          output       bn/act\n");
    for (int i = 0; i < n->nlayers; i++) {
        const orc_layer *L = &n->layers[i];
        if (L->kind == ORC_YOLO) {
            EMIT("%3d %8s class_num: %d ignore_thres: %3.2f [%d, %d] [%d, %d] [%d, %d]\n", i, KIND[L->kind], L->classes, L->thresh,
                 L->anchors[0][0], L->anchors[0][1], L->anchors[1][0], L->anchors[1][1], L->anchors[2][0], L->anchors[2][1]);
        } else if (L->kind == ORC_DROPOUT) {
            EMIT("%3d %8s %-38s -> %3dx%3dx%3d\n", i, KIND[L->kind], "", L->ow, L->oh, L->oc);
        } else if (L->kind == ORC_SHORTCUT || L->kind == ORC_ROUTE) {
            char deps[256] = "layers:";
            for (int k = 0; k < L->ndep; k++) { char t[16]; snprintf(t, sizeof t, " %d", L->dep[k]); strncat(deps, t, sizeof deps - strlen(deps) - 1); }
            EMIT("%3d %8s %-38s -> %3dx%3dx%3d\n", i, KIND[L->kind], deps, L->ow, L->oh, L->oc);
        } else {
            EMIT("%3d %8s %3d/%3d %2dx%2dx%3d   %d/%2d   %3dx%3dx%3d -> %3dx%3dx%3d  %d/%-6s\n", i, KIND[L->kind],
                 L->fn, L->groups, L->fs, L->fs, L->ic / L->groups, L->pad, L->stride, L->iw, L->ih, L->ic,
                 L->ow, L->oh, L->oc, L->batchnorm, (L->act >= 0 && L->act <= 2) ? ACT[L->act] : "unknown");
        }
    }
#undef EMIT
    return o;
}

/* bmpfile.c:42-69: 54-byte header, rows stored bottom-up, delivered top-down */
unsigned char *orc_bmp_load(const char *path, int *w, int *h)
{
    FILE *fp = fopen(path, "rb");
    unsigned char hdr[54];
    if (!fp) return NULL;
    if (fread(hdr, 1, 54, fp) != 54) { fclose(fp); return NULL; }
    uint32_t W, H;
    memcpy(&W, hdr + 18, 4); memcpy(&H, hdr + 22, 4);
    const int pitch = round_up((int)W * 3, 4);
    unsigned char *pix = malloc((size_t)pitch * H);
    if (pix) for (int y = (int)H - 1; y >= 0; y--) if (fread(pix + (size_t)y * pitch, pitch, 1, fp) != 1) break;
    fclose(fp);
    *w = (int)W; *h = (int)H;
    return pix;
}
