/*
 * gnnb_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see gnnb_oracle.h).
 *
 * CPU restatement of the reference hot path.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/).  The
 * arithmetic is scalar fp32 in the reference's loop order: neighbours in
 * CSR (stable COO) order, self term last where the reference adds it last.
 *
 * Where the reference's C++ and its PyTorch/PyG forward disagree, the PyTorch
 * forward is the parity target (SURVEY.md findings 5 and 7):
 *   - PNA std: sqrt(clamp(E[h^2]-E[h]^2, 1e-5)) then zeroed where
 *     <= sqrt(1e-5)   [GNNB_O_STD_PYG];   the library's Welford
 *     sqrt(var + 1e-5) is kept as GNNB_O_STD_HLS to cross-check oracle/_ref.
 *   - GELU is the exact erf form (nn.GELU default).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math, so the
 * fp32 operation order written here is the order executed).
 */
#include "gnnb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* gnnbuilder/gnn_builder_lib/gnn_builder_lib.h:1051-1083 */
void gnnb_oracle_degree_tables(const int32_t *coo, int n, int e, int32_t *in_deg, int32_t *out_deg)
{
    for (int i = 0; i < n; i++) {
        in_deg[i] = 0;
        out_deg[i] = 0;
    }
    for (int i = 0; i < e; i++) {
        int src = coo[2 * i + 0];
        int dst = coo[2 * i + 1];
        in_deg[dst]++;
        out_deg[src]++;
    }
}

/* gnnbuilder/gnn_builder_lib/gnn_builder_lib.h:1086-1124: exclusive prefix sum
 * of the in-degree, then a stable counting sort of sources by destination. */
void gnnb_oracle_neighbor_tables(const int32_t *coo, const int32_t *in_deg, int n, int e,
                                 int32_t *offsets, int32_t *neighbors)
{
    if (n <= 0)
        return;
    int32_t *cursor = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    offsets[0] = 0;
    cursor[0] = 0;
    for (int i = 1; i < n; i++) {
        int32_t c = offsets[i - 1] + in_deg[i - 1];
        offsets[i] = c;
        cursor[i] = c;
    }
    for (int i = 0; i < e; i++) {
        int src = coo[2 * i + 0];
        int dst = coo[2 * i + 1];
        neighbors[cursor[dst]] = src;
        cursor[dst]++;
    }
    free(cursor);
}

/* gnn_builder_lib.h:808-905 with BLOCK_SIZE_IN = BLOCK_SIZE_OUT = 1: the output
 * starts at the bias and every product is added to it one at a time. */
void gnnb_oracle_linear(const float *x, float *y, const float *W, const float *b, int in, int out)
{
    for (int o = 0; o < out; o++) {
        float acc = b ? b[o] : 0.0f;
        const float *w = W + (size_t)o * (size_t)in;
        for (int i = 0; i < in; i++) {
            float t = 0.0f;
            t += w[i] * x[i];
            acc += t;
        }
        y[o] = acc;
    }
}

/* gnn_builder_lib.h:363-375 (relu), :378-385 (gelu, erf), :420-425 (sigmoid),
 * :436-448 (tanh); selection by class name at templates/model.cpp.jinja:164-175
 * except that GELU follows nn.GELU (exact erf), SURVEY finding 7. */
static float act1(float x, int kind)
{
    switch (kind) {
    case GNNB_O_ACT_RELU:
        return x > 0.0f ? x : 0.0f;
    case GNNB_O_ACT_GELU:
        return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    case GNNB_O_ACT_SIGMOID:
        return 1.0f / (1.0f + expf(-x));
    case GNNB_O_ACT_TANH:
        return tanhf(x);
    default:
        return x;
    }
}

void gnnb_oracle_activation(float *x, int64_t count, int kind)
{
    for (int64_t i = 0; i < count; i++)
        x[i] = act1(x[i], kind);
}

/* gnn_builder_lib.h:1213-1289 (gcn_conv_agg) + :1291-1387 (gcn_conv):
 * agg_i = sum_j x_j / sqrt((1+d_i)(1+d_j)) + x_i / sqrt((1+d_i)^2), then linear. */
void gnnb_oracle_gcn_conv(int n, const float *x, float *out, const int32_t *offsets,
                          const int32_t *neighbors, const int32_t *in_deg, const float *W,
                          const float *b, int fin, int fout)
{
    float *agg = (float *)malloc(sizeof(float) * (size_t)(fin > 0 ? fin : 1));
    for (int node = 0; node < n; node++) {
        int deg = in_deg[node];
        for (int i = 0; i < fin; i++)
            agg[i] = 0.0f;
        for (int k = 0; k < deg; k++) {
            int j = neighbors[offsets[node] + k];
            float di = 1.0f + (float)deg;
            float dj = 1.0f + (float)in_deg[j];
            float s = 1.0f / sqrtf(di * dj);
            for (int i = 0; i < fin; i++)
                agg[i] += x[(size_t)j * fin + i] * s;
        }
        float di = 1.0f + (float)deg;
        float sself = 1.0f / sqrtf(di * di);
        for (int i = 0; i < fin; i++)
            agg[i] += x[(size_t)node * fin + i] * sself;
        gnnb_oracle_linear(agg, out + (size_t)node * fout, W, b, fin, fout);
    }
    free(agg);
}

/* gnn_builder_lib.h:1389-1437 (gin_conv_agg) + :1440-1549 (gin_conv):
 * z = sum_j x_j + x_i (1+eps); out = W1 relu(W0 z + b0) + b1. */
void gnnb_oracle_gin_conv(int n, const float *x, float *out, const int32_t *offsets,
                          const int32_t *neighbors, const int32_t *in_deg, const float *W0,
                          const float *b0, const float *W1, const float *b1, float eps, int fin,
                          int hidden, int fout)
{
    float *z = (float *)malloc(sizeof(float) * (size_t)(fin > 0 ? fin : 1));
    float *h = (float *)malloc(sizeof(float) * (size_t)(hidden > 0 ? hidden : 1));
    for (int node = 0; node < n; node++) {
        for (int i = 0; i < fin; i++)
            z[i] = 0.0f;
        for (int k = 0; k < in_deg[node]; k++) {
            int j = neighbors[offsets[node] + k];
            for (int i = 0; i < fin; i++)
                z[i] += x[(size_t)j * fin + i];
        }
        for (int i = 0; i < fin; i++)
            z[i] = z[i] + x[(size_t)node * fin + i] * (1.0f + eps);
        gnnb_oracle_linear(z, h, W0, b0, fin, hidden);
        gnnb_oracle_activation(h, hidden, GNNB_O_ACT_RELU);
        gnnb_oracle_linear(h, out + (size_t)node * fout, W1, b1, hidden, fout);
    }
    free(z);
    free(h);
}

/* gnn_builder_lib.h:2161-2209 (sage_conv_agg, running sum / count) +
 * :2211-2341 (sage_conv): out = Wl mean_j x_j + bl + Wr x_i; empty mean = 0. */
void gnnb_oracle_sage_conv(int n, const float *x, float *out, const int32_t *offsets,
                           const int32_t *neighbors, const int32_t *in_deg, const float *Wl,
                           const float *bl, const float *Wr, int fin, int fout)
{
    float *m = (float *)malloc(sizeof(float) * (size_t)(fin > 0 ? fin : 1));
    float *t = (float *)malloc(sizeof(float) * (size_t)(fout > 0 ? fout : 1));
    for (int node = 0; node < n; node++) {
        int deg = in_deg[node];
        for (int i = 0; i < fin; i++)
            m[i] = 0.0f;
        for (int k = 0; k < deg; k++) {
            int j = neighbors[offsets[node] + k];
            for (int i = 0; i < fin; i++)
                m[i] += x[(size_t)j * fin + i];
        }
        if (deg > 0)
            for (int i = 0; i < fin; i++)
                m[i] = m[i] / (float)deg;
        float *o = out + (size_t)node * fout;
        gnnb_oracle_linear(m, o, Wl, bl, fin, fout);
        gnnb_oracle_linear(x + (size_t)node * fin, t, Wr, NULL, fin, fout);
        for (int i = 0; i < fout; i++)
            o[i] = o[i] + t[i];
    }
    free(m);
    free(t);
}

/* gnn_builder_lib.h:1750-1834 (pna_conv_agg), :1836-1876 (concat order),
 * :1891-2157 (pna_conv); wrapper gnnbuilder/models.py:209-240.
 * Per edge h_ij = Wpre [x_i || x_j] + bpre (destination first, lib:1801-1802);
 * max/min/mean/std over j; scalers with d = max(deg,1); 13F concat -> Wpost
 * -> Wlin (no nonlinearity between).  Empty neighbourhood: every aggregate 0
 * (lib:739,776 defaults; matches PyG's fill value). */
void gnnb_oracle_pna_conv(int n, const float *x, float *out, const int32_t *offsets,
                          const int32_t *neighbors, const int32_t *in_deg, const float *Wpre,
                          const float *bpre, const float *Wpost, const float *bpost,
                          const float *Wlin, const float *blin, float delta, int std_mode, int fin,
                          int fout)
{
    size_t f = (size_t)(fin > 0 ? fin : 1);
    float *cat2 = (float *)malloc(sizeof(float) * 2 * f);
    float *h = (float *)malloc(sizeof(float) * f);
    float *vmax = (float *)malloc(sizeof(float) * f);
    float *vmin = (float *)malloc(sizeof(float) * f);
    float *s1 = (float *)malloc(sizeof(float) * f);
    float *s2 = (float *)malloc(sizeof(float) * f);
    float *wmean = (float *)malloc(sizeof(float) * f); /* Welford (HLS mode) */
    float *wm2 = (float *)malloc(sizeof(float) * f);
    float *cat13 = (float *)malloc(sizeof(float) * 13 * f);
    float *hid = (float *)malloc(sizeof(float) * (size_t)(fout > 0 ? fout : 1));
    const float thr = sqrtf(1e-5f);

    for (int node = 0; node < n; node++) {
        int deg = in_deg[node];
        const float *xi = x + (size_t)node * fin;
        for (int i = 0; i < fin; i++) {
            vmax[i] = 0.0f;
            vmin[i] = 0.0f;
            s1[i] = 0.0f;
            s2[i] = 0.0f;
            wmean[i] = 0.0f;
            wm2[i] = 0.0f;
        }
        for (int k = 0; k < deg; k++) {
            int j = neighbors[offsets[node] + k];
            const float *xj = x + (size_t)j * fin;
            for (int i = 0; i < fin; i++) {
                cat2[i] = xi[i];
                cat2[fin + i] = xj[i];
            }
            gnnb_oracle_linear(cat2, h, Wpre, bpre, 2 * fin, fin);
            for (int i = 0; i < fin; i++) {
                float v = h[i];
                if (k == 0) {
                    vmax[i] = v;
                    vmin[i] = v;
                } else {
                    if (v > vmax[i])
                        vmax[i] = v;
                    if (v < vmin[i])
                        vmin[i] = v;
                }
                s1[i] += v;
                s2[i] += v * v;
                /* lib:689-696 Welford update */
                float d0 = v - wmean[i];
                wmean[i] += d0 / (float)(k + 1);
                wm2[i] += d0 * (v - wmean[i]);
            }
        }
        int dcl = deg < 1 ? 1 : deg;
        float logd = logf((float)(dcl + 1));
        float amp = logd / delta;
        float att = delta / logd;
        for (int i = 0; i < fin; i++) {
            float mean = 0.0f, sd = 0.0f;
            if (deg > 0) {
                mean = s1[i] / (float)deg;
                if (std_mode == GNNB_O_STD_HLS) {
                    /* lib:698-704 */
                    sd = sqrtf(wm2[i] / (float)deg + 1e-5f);
                } else {
                    /* PyG StdAggregation as pinned by tb_pna_output.bin (SURVEY finding 5) */
                    float mean2 = s2[i] / (float)deg;
                    float var = mean2 - mean * mean;
                    if (var < 1e-5f)
                        var = 1e-5f;
                    sd = sqrtf(var);
                    if (sd <= thr)
                        sd = 0.0f;
                }
            } else if (std_mode == GNNB_O_STD_HLS) {
                /* the library divides 0/0 here (lib:702); keep PyG's 0 instead of NaN */
                sd = 0.0f;
            }
            float a4[4] = {vmax[i], vmin[i], mean, sd};
            cat13[i] = xi[i];
            for (int q = 0; q < 4; q++) {
                cat13[(size_t)(1 + q) * fin + i] = a4[q];
                cat13[(size_t)(5 + q) * fin + i] = amp * a4[q];
                cat13[(size_t)(9 + q) * fin + i] = att * a4[q];
            }
        }
        gnnb_oracle_linear(cat13, hid, Wpost, bpost, 13 * fin, fout);
        gnnb_oracle_linear(hid, out + (size_t)node * fout, Wlin, blin, fout, fout);
    }
    free(cat2);
    free(h);
    free(vmax);
    free(vmin);
    free(s1);
    free(s2);
    free(wmean);
    free(wm2);
    free(cat13);
    free(hid);
}

/* gnn_builder_lib.h:1126-1166 (compute_neighbor_and_edge_index_tables): the neighbour table plus, per CSR slot,
 * the COO row of its edge. */
void gnnb_oracle_neighbor_edge_tables(const int32_t *coo, const int32_t *in_deg, int n, int e, int32_t *offsets,
                                      int32_t *neighbors, int32_t *edge_index)
{
    int32_t *cursor = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    if (n > 0) {
        offsets[0] = 0;
        cursor[0] = 0;
    }
    for (int i = 1; i < n; i++) {
        offsets[i] = offsets[i - 1] + in_deg[i - 1];
        cursor[i] = offsets[i];
    }
    for (int i = 0; i < e; i++) {
        int src = coo[2 * i], dst = coo[2 * i + 1];
        neighbors[cursor[dst]] = src;
        edge_index[cursor[dst]] = i;
        cursor[dst]++;
    }
    free(cursor);
}

/* gnn_builder_lib.h:1555-1625 (gine_conv_agg) + :1640-1742 (gine_conv):
 * z = sum_j relu(x_j + (We e_ij + be)) + x_i (1+eps); out = W1 relu(W0 z + b0) + b1. */
void gnnb_oracle_gine_conv(int n, const float *x, const float *edge_attr, float *out, const int32_t *offsets,
                           const int32_t *neighbors, const int32_t *edge_index, const int32_t *in_deg,
                           const float *We, const float *be, const float *W0, const float *b0, const float *W1,
                           const float *b1, float eps, int fin, int edge_dim, int hidden, int fout)
{
    float *z = (float *)malloc(sizeof(float) * (size_t)(fin > 0 ? fin : 1));
    float *pe = (float *)malloc(sizeof(float) * (size_t)(fin > 0 ? fin : 1));
    float *h = (float *)malloc(sizeof(float) * (size_t)(hidden > 0 ? hidden : 1));
    for (int node = 0; node < n; node++) {
        for (int i = 0; i < fin; i++)
            z[i] = 0.0f;
        for (int k = 0; k < in_deg[node]; k++) {
            int j = neighbors[offsets[node] + k];
            int e = edge_index[offsets[node] + k];
            gnnb_oracle_linear(edge_attr + (size_t)e * edge_dim, pe, We, be, edge_dim, fin);
            for (int i = 0; i < fin; i++) {
                float m = x[(size_t)j * fin + i] + pe[i];
                z[i] += m > 0.0f ? m : 0.0f;
            }
        }
        for (int i = 0; i < fin; i++)
            z[i] = z[i] + x[(size_t)node * fin + i] * (1.0f + eps);
        gnnb_oracle_linear(z, h, W0, b0, fin, hidden);
        for (int i = 0; i < hidden; i++)
            h[i] = h[i] > 0.0f ? h[i] : 0.0f;
        gnnb_oracle_linear(h, out + (size_t)node * fout, W1, b1, hidden, fout);
    }
    free(z);
    free(pe);
    free(h);
}

/* gnn_builder_lib.h:2501-2634 (simple_conv, aggregation "sum") */
void gnnb_oracle_simple_conv(int n, const float *x, float *out, const int32_t *offsets,
                             const int32_t *neighbors, const int32_t *in_deg, int f)
{
    for (int node = 0; node < n; node++) {
        float *o = out + (size_t)node * f;
        for (int i = 0; i < f; i++)
            o[i] = 0.0f;
        for (int k = 0; k < in_deg[node]; k++) {
            int j = neighbors[offsets[node] + k];
            for (int i = 0; i < f; i++)
                o[i] += x[(size_t)j * f + i];
        }
    }
}

/* gnn_builder_lib.h:2350-2499 (lg_conv): sum_j x_j / sqrt(d_i d_j), no self
 * term; a zero degree product contributes 0 (PyG masks the inf). */
void gnnb_oracle_lg_conv(int n, const float *x, float *out, const int32_t *offsets,
                         const int32_t *neighbors, const int32_t *in_deg, int f)
{
    for (int node = 0; node < n; node++) {
        float *o = out + (size_t)node * f;
        for (int i = 0; i < f; i++)
            o[i] = 0.0f;
        for (int k = 0; k < in_deg[node]; k++) {
            int j = neighbors[offsets[node] + k];
            int prod = in_deg[node] * in_deg[j];
            float s = prod > 0 ? 1.0f / sqrtf((float)prod) : 0.0f;
            for (int i = 0; i < f; i++)
                o[i] += x[(size_t)j * f + i] * s;
        }
    }
}

/* gnn_builder_lib.h:2709-2739 (add), :2741-2771 (mean = running sum / count),
 * :2773-2803 (max, first sample initialises; empty -> 0) */
void gnnb_oracle_global_pool(const float *x, int n, int d, int kind, float *out)
{
    for (int j = 0; j < d; j++) {
        float acc = 0.0f;
        for (int i = 0; i < n; i++) {
            float v = x[(size_t)i * d + j];
            if (kind == GNNB_O_POOL_MAX)
                acc = (i == 0 || v > acc) ? v : acc;
            else
                acc += v;
        }
        if (kind == GNNB_O_POOL_MEAN && n > 0)
            acc = acc / (float)n;
        out[j] = acc;
    }
}

static int conv_slots(int conv_type)
{
    switch (conv_type) {
    case GNNB_O_CONV_GCN:
        return 2;
    case GNNB_O_CONV_GIN:
        return 4;
    case GNNB_O_CONV_SAGE:
        return 3;
    case GNNB_O_CONV_PNA:
        return 6;
    default:
        return -1;
    }
}

int gnnb_oracle_num_params(const gnnb_oracle_desc *d)
{
    int s = conv_slots(d->conv_type);
    if (s < 0)
        return -1;
    return s * d->num_layers + 2 * d->mlp_num_linear;
}

/* layer dims: gnnbuilder/models.py:519-549 */
static void layer_dims(const gnnb_oracle_desc *d, int layer, int *fin, int *fout)
{
    int L = d->num_layers;
    if (L == 1) {
        *fin = d->in_dim;
        *fout = d->out_dim;
    } else if (layer == 0) {
        *fin = d->in_dim;
        *fout = d->hidden_dim;
    } else if (layer == L - 1) {
        *fin = d->hidden_dim;
        *fout = d->out_dim;
    } else {
        *fin = d->hidden_dim;
        *fout = d->hidden_dim;
    }
}

/* ap_fixed<W, I, AP_TRN, AP_WRAP> grid (code_gen.py:39-52, model.h.jinja:41-45): truncate towards minus infinity
 * to a multiple of 2^-(W-I), wrap into [-2^(I-1), 2^(I-1)). */
static void quantize_fpx(float *v, size_t n, int W, int I)
{
    if (W <= 0)
        return;
    const float inv = ldexpf(1.0f, W - I), step = ldexpf(1.0f, -(W - I));
    const float span = ldexpf(1.0f, I), half = ldexpf(1.0f, I - 1);
    for (size_t i = 0; i < n; i++) {
        float q = floorf(v[i] * inv) * step;
        v[i] = q - span * floorf((q + half) / span);
    }
}

/* gnnbuilder/models.py:551-575 (GNNModel.forward), generated counterpart
 * templates/model.cpp.jinja:151-359 (conv stack, skip :304-311, activation
 * :313-322), :413-449 (pool concat), :454-530 (MLP head, models.py:398-430). */
int gnnb_oracle_forward(const gnnb_oracle_desc *d, const float *const *params, const float *x,
                        const int32_t *coo, int n, int e, float *out)
{
    int slots = conv_slots(d->conv_type);
    if (slots < 0 || d->num_layers < 0 || d->num_pools < 1 || d->num_pools > 3 ||
        d->mlp_num_linear < 1)
        return -1;
    if (d->num_layers == 0 && d->in_dim != d->out_dim)
        return -2; /* models.py:512-518 */

    size_t nn = (size_t)(n > 0 ? n : 1), ee = (size_t)(e > 0 ? e : 1);
    /* GCN, PyG semantics: explicit self loops are not edges (gcn_norm -> add_remaining_self_loops keeps exactly
     * one self loop per node); the reference C++ would count them on top of its self term (HLS mode). */
    int32_t *noloop = NULL;
    if (d->conv_type == GNNB_O_CONV_GCN && d->gcn_self_loop_mode == GNNB_O_SELF_LOOPS_PYG) {
        noloop = (int32_t *)malloc(sizeof(int32_t) * 2 * ee);
        int m = 0;
        for (int i = 0; i < e; i++)
            if (coo[2 * i] != coo[2 * i + 1]) {
                noloop[2 * m] = coo[2 * i];
                noloop[2 * m + 1] = coo[2 * i + 1];
                m++;
            }
        coo = noloop;
        e = m;
    }
    int32_t *in_deg = (int32_t *)malloc(sizeof(int32_t) * nn);
    int32_t *out_deg = (int32_t *)malloc(sizeof(int32_t) * nn);
    int32_t *offsets = (int32_t *)malloc(sizeof(int32_t) * nn);
    int32_t *nbrs = (int32_t *)malloc(sizeof(int32_t) * ee);
    gnnb_oracle_degree_tables(coo, n, e, in_deg, out_deg);
    gnnb_oracle_neighbor_tables(coo, in_deg, n, e, offsets, nbrs);

    int maxd = d->in_dim;
    if (d->hidden_dim > maxd)
        maxd = d->hidden_dim;
    if (d->out_dim > maxd)
        maxd = d->out_dim;
    float *cur = (float *)malloc(sizeof(float) * nn * (size_t)maxd);
    float *nxt = (float *)malloc(sizeof(float) * nn * (size_t)maxd);
    memcpy(cur, x, sizeof(float) * (size_t)n * (size_t)d->in_dim);
    quantize_fpx(cur, (size_t)n * (size_t)d->in_dim, d->fpx_w, d->fpx_i); /* F_TYPE inputs */
    int width = d->in_dim;

    /* fixed-point emulation: W_TYPE weights live on the grid too -- quantised copies, same order */
    const float *const *p = params;
    float **qparams = NULL;
    int nparams = 0;
    if (d->fpx_w > 0) {
        nparams = gnnb_oracle_num_params(d);
        qparams = (float **)malloc(sizeof(float *) * (size_t)nparams);
        int pi = 0;
        for (int l = 0; l < d->num_layers; l++) {
            int fi, fo;
            layer_dims(d, l, &fi, &fo);
            size_t sz[6];
            int ns = 0;
            switch (d->conv_type) {
            case GNNB_O_CONV_GCN: sz[0] = (size_t)fo * fi; sz[1] = fo; ns = 2; break;
            case GNNB_O_CONV_GIN: sz[0] = (size_t)fo * fi; sz[1] = fo; sz[2] = (size_t)fo * fo; sz[3] = fo; ns = 4; break;
            case GNNB_O_CONV_SAGE: sz[0] = (size_t)fo * fi; sz[1] = fo; sz[2] = (size_t)fo * fi; ns = 3; break;
            default: sz[0] = (size_t)fi * 2 * fi; sz[1] = fi; sz[2] = (size_t)fo * 13 * fi; sz[3] = fo; sz[4] = (size_t)fo * fo; sz[5] = fo; ns = 6; break;
            }
            for (int k = 0; k < ns; k++, pi++) {
                qparams[pi] = (float *)malloc(sizeof(float) * sz[k]);
                memcpy(qparams[pi], params[pi], sizeof(float) * sz[k]);
                quantize_fpx(qparams[pi], sz[k], d->fpx_w, d->fpx_i);
            }
        }
        int din = d->num_pools * (d->num_layers == 0 ? d->in_dim : d->out_dim);
        for (int l = 0; l < d->mlp_num_linear; l++) {
            int dout = (l == d->mlp_num_linear - 1) ? d->mlp_out : d->mlp_hidden;
            size_t sw = (size_t)din * dout;
            qparams[pi] = (float *)malloc(sizeof(float) * sw);
            memcpy(qparams[pi], params[pi], sizeof(float) * sw);
            quantize_fpx(qparams[pi], sw, d->fpx_w, d->fpx_i);
            pi++;
            qparams[pi] = (float *)malloc(sizeof(float) * (size_t)dout);
            memcpy(qparams[pi], params[pi], sizeof(float) * (size_t)dout);
            quantize_fpx(qparams[pi], (size_t)dout, d->fpx_w, d->fpx_i);
            pi++;
            din = dout;
        }
        p = (const float *const *)qparams;
    }
    for (int l = 0; l < d->num_layers; l++) {
        int fin, fout;
        layer_dims(d, l, &fin, &fout);
        switch (d->conv_type) {
        case GNNB_O_CONV_GCN:
            gnnb_oracle_gcn_conv(n, cur, nxt, offsets, nbrs, in_deg, p[0], p[1], fin, fout);
            break;
        case GNNB_O_CONV_GIN:
            /* hidden = out_channels (models.py:90, SURVEY finding 6) */
            gnnb_oracle_gin_conv(n, cur, nxt, offsets, nbrs, in_deg, p[0], p[1], p[2], p[3],
                                 d->gin_eps, fin, fout, fout);
            break;
        case GNNB_O_CONV_SAGE:
            gnnb_oracle_sage_conv(n, cur, nxt, offsets, nbrs, in_deg, p[0], p[1], p[2], fin, fout);
            break;
        case GNNB_O_CONV_PNA:
            gnnb_oracle_pna_conv(n, cur, nxt, offsets, nbrs, in_deg, p[0], p[1], p[2], p[3], p[4],
                                 p[5], d->pna_delta, d->pna_std_mode, fin, fout);
            break;
        }
        p += slots;
        /* skip on middle layers only: models.py:562-564 */
        if (d->skip && l != 0 && l != d->num_layers - 1) {
            for (size_t i = 0; i < (size_t)n * (size_t)fout; i++)
                nxt[i] = nxt[i] + cur[i];
        }
        gnnb_oracle_activation(nxt, (int64_t)n * fout, d->activation);
        quantize_fpx(nxt, (size_t)n * (size_t)fout, d->fpx_w, d->fpx_i);
        float *t = cur;
        cur = nxt;
        nxt = t;
        width = fout;
    }

    int pooled_dim = d->num_pools * width;
    float *pooled = (float *)malloc(sizeof(float) * (size_t)pooled_dim);
    for (int k = 0; k < d->num_pools; k++)
        gnnb_oracle_global_pool(cur, n, width, d->pools[k], pooled + (size_t)k * width);
    quantize_fpx(pooled, (size_t)pooled_dim, d->fpx_w, d->fpx_i);

    int hd = d->mlp_hidden > d->mlp_out ? d->mlp_hidden : d->mlp_out;
    if (pooled_dim > hd)
        hd = pooled_dim;
    float *a = (float *)malloc(sizeof(float) * (size_t)hd);
    float *b = (float *)malloc(sizeof(float) * (size_t)hd);
    memcpy(a, pooled, sizeof(float) * (size_t)pooled_dim);
    int din = pooled_dim;
    for (int l = 0; l < d->mlp_num_linear; l++) {
        int last = (l == d->mlp_num_linear - 1);
        int dout = last ? d->mlp_out : d->mlp_hidden;
        gnnb_oracle_linear(a, b, p[0], p[1], din, dout);
        p += 2;
        if (!last)
            gnnb_oracle_activation(b, dout, d->mlp_activation);
        quantize_fpx(b, (size_t)dout, d->fpx_w, d->fpx_i);
        float *t = a;
        a = b;
        b = t;
        din = dout;
    }
    memcpy(out, a, sizeof(float) * (size_t)d->mlp_out);
    if (d->output_activation == 1 || d->output_activation == 2) { /* models.py:572-573: module(dim=-1) */
        float mx = out[0], sum = 0.0f;
        for (int i = 1; i < d->mlp_out; i++)
            mx = out[i] > mx ? out[i] : mx;
        for (int i = 0; i < d->mlp_out; i++)
            sum += expf(out[i] - mx);
        for (int i = 0; i < d->mlp_out; i++)
            out[i] = d->output_activation == 1 ? expf(out[i] - mx) / sum : (out[i] - mx) - logf(sum);
    }

    free(a);
    free(b);
    free(pooled);
    free(cur);
    free(nxt);
    free(in_deg);
    free(out_deg);
    free(offsets);
    free(nbrs);
    free(noloop);
    if (qparams) {
        for (int i = 0; i < nparams; i++)
            free(qparams[i]);
        free(qparams);
    }
    return 0;
}

int gnnb_oracle_forward_batched(const gnnb_oracle_desc *d, const float *const *params,
                                const float *x, const int32_t *coo, const int32_t *node_ptr,
                                const int32_t *edge_ptr, int num_graphs, float *out)
{
    for (int g = 0; g < num_graphs; g++) {
        int n0 = node_ptr[g], n = node_ptr[g + 1] - n0;
        int e0 = edge_ptr[g], e = edge_ptr[g + 1] - e0;
        int32_t *local = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(e > 0 ? e : 1));
        for (int i = 0; i < e; i++) {
            int s = coo[2 * (size_t)(e0 + i) + 0] - n0;
            int t = coo[2 * (size_t)(e0 + i) + 1] - n0;
            if (s < 0 || s >= n || t < 0 || t >= n) {
                free(local);
                return -3; /* edge leaves its graph */
            }
            local[2 * i + 0] = s;
            local[2 * i + 1] = t;
        }
        int rc = gnnb_oracle_forward(d, params, x + (size_t)n0 * d->in_dim, local, n, e,
                                     out + (size_t)g * d->mlp_out);
        free(local);
        if (rc != 0)
            return rc;
    }
    return 0;
}
