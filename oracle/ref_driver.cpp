// ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" driver over the REFERENCE's own header-only kernel library,
// compiled in place from /root/reference (never copied into this repo):
//     gnnbuilder/gnn_builder_lib/gnn_builder_lib.h
// The library is configured the way its generated includer configures it in
// float mode (templates/model.h.jinja:18-36): F_TYPE/W_TYPE = float and the
// m_* math macros mapped to <cmath>.  Those macros are the library's own
// configuration interface, not stand-ins for a missing dependency; the
// fixed-point (Vitis ap_fixed) mode is unbuildable here and not attempted.
//
// The output (oracle/_ref/libgnnb_ref.so) is used to (1) validate the C
// restatement in gnnb_oracle.c and (2) serve as bench.py's cpu_baseline of
// kind "reference".  It exists only where /root/reference exists at build time;
// the GPU box receives the prebuilt .so.
//
// The reference kernels are C++ templates over static sizes, so this driver
// instantiates them for a fixed table of (F_in, F_out) pairs and dispatches at
// run time; an unlisted size returns -1 and callers skip the cross-check.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#define __FLOATING_POINT_MODEL__ 1 // templates/model.h.jinja:3-5 (float_or_fixed == "float")
typedef float F_TYPE;
typedef float W_TYPE;
#define m_sqrt(x) (std::sqrt(x))
#define m_rsqrt(x) (F_TYPE(1.0) / std::sqrt(x))
#define m_recip(x) (F_TYPE(1.0) / x)
#define m_erf(x) (std::erf(x))
#define m_tanh(x) (std::tanh(x))
#define m_pow(x, y) (std::pow(x, y))
#define m_exp(x) (std::exp(x))
#define m_log(x) (std::log(x))
#define m_abs(x) (std::abs(x))
#define m_sin(x) (std::sin(x))
#define m_cos(x) (std::cos(x))
#define m_pi() ((float)3.14159265358979323846)
#define m_signbit(x) (std::signbit(x))

#include "gnn_builder_lib.h" // found via -I/root/reference/gnnbuilder/gnn_builder_lib

#define REF_MAX_NODES 600
#define REF_MAX_EDGES 1500

// (F_in, F_out) pairs the convs are instantiated for.
#define REF_CONV_PAIRS(X) \
    X(8, 8)               \
    X(9, 16)              \
    X(11, 16)             \
    X(16, 16)             \
    X(16, 8)              \
    X(9, 64)              \
    X(64, 64)             \
    X(9, 128)             \
    X(11, 128)            \
    X(128, 128)           \
    X(128, 64)            \
    X(9, 256)             \
    X(256, 256)

// (in, out) pairs for the MLP head's linear layers.
#define REF_LINEAR_PAIRS(X) \
    X(8, 8)                 \
    X(16, 8)                \
    X(8, 3)                 \
    X(16, 64)               \
    X(24, 16)               \
    X(32, 16)               \
    X(48, 16)               \
    X(16, 16)               \
    X(16, 5)                \
    X(16, 1)                \
    X(48, 64)               \
    X(64, 64)               \
    X(64, 19)               \
    X(64, 1)                \
    X(64, 2)                \
    X(128, 64)              \
    X(192, 64)              \
    X(384, 64)              \
    X(256, 64)              \
    X(768, 64)

#define REF_POOL_DIMS(X) \
    X(8)                 \
    X(16)                \
    X(64)                \
    X(128)               \
    X(256)

typedef int edge_arr_t[2];

extern "C" {

int gnnb_ref_max_nodes() { return REF_MAX_NODES; }
int gnnb_ref_max_edges() { return REF_MAX_EDGES; }

// gnn_builder_lib.h:1051-1124
int gnnb_ref_tables(const int32_t *coo, int n, int e, int32_t *in_deg, int32_t *out_deg,
                    int32_t *offsets, int32_t *neighbors)
{
    if (n > REF_MAX_NODES || e > REF_MAX_EDGES)
        return -1;
    compute_degree_tables<REF_MAX_NODES, REF_MAX_EDGES>((edge_arr_t *)coo, in_deg, out_deg, n, e);
    compute_neighbor_tables<REF_MAX_NODES, REF_MAX_EDGES>((edge_arr_t *)coo, in_deg, out_deg,
                                                          offsets, neighbors, n, e);
    return 0;
}

int gnnb_ref_gcn_conv(int n, int e, const float *x, float *out, const int32_t *coo,
                      const int32_t *offsets, const int32_t *neighbors, const int32_t *in_deg,
                      const int32_t *out_deg, const float *W, const float *b, int fin, int fout)
{
    if (n > REF_MAX_NODES || e > REF_MAX_EDGES)
        return -1;
#define X(FI, FO)                                                                              \
    if (fin == FI && fout == FO) {                                                             \
        gcn_conv<REF_MAX_NODES, REF_MAX_EDGES, FI, FO, float>(                                 \
            n, e, (float(*)[FI])x, (float(*)[FO])out, (edge_arr_t *)coo, (int *)offsets,       \
            (int *)neighbors, (int *)in_deg, (int *)out_deg, (float(*)[FI])W, (float *)b);     \
        return 0;                                                                              \
    }
    REF_CONV_PAIRS(X)
#undef X
    return -1;
}

int gnnb_ref_gin_conv(int n, int e, const float *x, float *out, const int32_t *coo,
                      const int32_t *offsets, const int32_t *neighbors, const int32_t *in_deg,
                      const int32_t *out_deg, const float *W0, const float *b0, const float *W1,
                      const float *b1, float eps, int fin, int fout)
{
    if (n > REF_MAX_NODES || e > REF_MAX_EDGES)
        return -1;
    // hidden = out_channels (gnnbuilder/models.py:90)
#define X(FI, FO)                                                                              \
    if (fin == FI && fout == FO) {                                                             \
        gin_conv<REF_MAX_NODES, REF_MAX_EDGES, FI, FO, FO, float>(                             \
            n, e, (float(*)[FI])x, (float(*)[FO])out, (edge_arr_t *)coo, (int *)offsets,       \
            (int *)neighbors, (int *)in_deg, (int *)out_deg, (float(*)[FI])W0, (float *)b0,    \
            (float(*)[FO])W1, (float *)b1, eps);                                               \
        return 0;                                                                              \
    }
    REF_CONV_PAIRS(X)
#undef X
    return -1;
}

int gnnb_ref_sage_conv(int n, int e, const float *x, float *out, const int32_t *coo,
                       const int32_t *offsets, const int32_t *neighbors, const int32_t *in_deg,
                       const int32_t *out_deg, const float *Wl, const float *bl, const float *Wr,
                       int fin, int fout)
{
    if (n > REF_MAX_NODES || e > REF_MAX_EDGES)
        return -1;
#define X(FI, FO)                                                                              \
    if (fin == FI && fout == FO) {                                                             \
        sage_conv<REF_MAX_NODES, REF_MAX_EDGES, FI, FO, float>(                                \
            n, e, (float(*)[FI])x, (float(*)[FO])out, (edge_arr_t *)coo, (int *)offsets,       \
            (int *)neighbors, (int *)in_deg, (int *)out_deg, (float(*)[FI])Wl, (float *)bl,    \
            (float(*)[FI])Wr);                                                                 \
        return 0;                                                                              \
    }
    REF_CONV_PAIRS(X)
#undef X
    return -1;
}

// NOTE: the library's std is sqrt(var_welford + 1e-5) (gnn_builder_lib.h:698-704), which is
// NOT what the PyTorch/PyG forward computes; compare against GNNB_O_STD_HLS.
int gnnb_ref_pna_conv(int n, int e, const float *x, float *out, const int32_t *coo,
                      const int32_t *offsets, const int32_t *neighbors, const int32_t *in_deg,
                      const int32_t *out_deg, const float *Wpre, const float *bpre,
                      const float *Wpost, const float *bpost, const float *Wlin,
                      const float *blin, float delta, int fin, int fout)
{
    if (n > REF_MAX_NODES || e > REF_MAX_EDGES)
        return -1;
#define X(FI, FO)                                                                              \
    if (fin == FI && fout == FO) {                                                             \
        pna_conv<REF_MAX_NODES, REF_MAX_EDGES, FI, FO, 2 * FI, FI, 13 * FI, FO, float>(        \
            n, e, (float(*)[FI])x, (float(*)[FO])out, (edge_arr_t *)coo, (int *)offsets,       \
            (int *)neighbors, (int *)in_deg, (int *)out_deg, (float(*)[2 * FI])Wpre,           \
            (float *)bpre, (float(*)[13 * FI])Wpost, (float *)bpost, (float(*)[FO])Wlin,       \
            (float *)blin, delta);                                                             \
        return 0;                                                                              \
    }
    REF_CONV_PAIRS(X)
#undef X
    return -1;
}

int gnnb_ref_linear(const float *x, float *y, const float *W, const float *b, int in, int out)
{
#define X(I, O)                                                                    \
    if (in == I && out == O) {                                                     \
        linear<I, O, 1, 1, float>((float *)x, y, (float(*)[I])W, (float *)b);      \
        return 0;                                                                  \
    }
    REF_LINEAR_PAIRS(X)
#undef X
    return -1;
}

// kind: 0 add, 1 mean, 2 max (gnn_builder_lib.h:2709-2803)
int gnnb_ref_global_pool(const float *x, int n, int d, int kind, float *out)
{
    if (n > REF_MAX_NODES)
        return -1;
#define X(D)                                                                                       \
    if (d == D) {                                                                                  \
        if (kind == 0)                                                                             \
            global_add_pool<REF_MAX_NODES, REF_MAX_EDGES, D, float>(n, 0, (float(*)[D])x, out);    \
        else if (kind == 1)                                                                        \
            global_mean_pool<REF_MAX_NODES, REF_MAX_EDGES, D, float>(n, 0, (float(*)[D])x, out);   \
        else                                                                                       \
            global_max_pool<REF_MAX_NODES, REF_MAX_EDGES, D, float>(n, 0, (float(*)[D])x, out);    \
        return 0;                                                                                  \
    }
    REF_POOL_DIMS(X)
#undef X
    return -1;
}

// kind: 0 relu, 1 gelu (erf, lib:378-385), 2 sigmoid, 3 tanh
void gnnb_ref_activation(float *x, long count, int kind)
{
    for (long i = 0; i < count; i++) {
        switch (kind) {
        case 0: x[i] = activation_relu<float>(x[i]); break;
        case 1: x[i] = activation_gelu<float>(x[i]); break;
        case 2: x[i] = activation_sigmoid<float>(x[i]); break;
        case 3: x[i] = activation_tanh<float>(x[i]); break;
        default: break;
        }
    }
}


// ---------------------------------------------------------------------------------------
// Whole model on ONE graph, composed from the reference's compiled kernels in the order its
// generated top uses (templates/model.cpp.jinja:737-765): tables -> per layer conv, skip on
// middle layers (:304-311), activation (:313-322) -> pooling concat (:440-448) -> MLP head
// (:454-530).  `desc` has the layout of gnnb_oracle_desc (oracle/gnnb_oracle.h); params in the
// canonical order documented there.  Returns -1 when a size has no instantiation.
struct ref_desc {
    int32_t conv_type, num_layers, in_dim, hidden_dim, out_dim, activation, skip, num_pools;
    int32_t pools[3];
    int32_t mlp_num_linear, mlp_hidden, mlp_out, mlp_activation;
    float gin_eps, pna_delta;
    int32_t pna_std_mode;
};

int gnnb_ref_forward(const ref_desc *d, const float *const *params, const float *x,
                     const int32_t *coo, int n, int e, float *out)
{
    if (n > REF_MAX_NODES || e > REF_MAX_EDGES)
        return -1;
    static int32_t in_deg[REF_MAX_NODES], out_deg[REF_MAX_NODES], offsets[REF_MAX_NODES],
        nbrs[REF_MAX_EDGES];
    static float bufa[REF_MAX_NODES * 256], bufb[REF_MAX_NODES * 256];
    static float pooled[3 * 256], h0[1024], h1[1024];
    if (d->in_dim > 256 || d->hidden_dim > 256 || d->out_dim > 256 || d->mlp_hidden > 1024 ||
        d->mlp_out > 1024)
        return -1;
    if (gnnb_ref_tables(coo, n, e, in_deg, out_deg, offsets, nbrs) != 0)
        return -1;
    float *cur = bufa, *nxt = bufb;
    memcpy(cur, x, sizeof(float) * (size_t)n * d->in_dim);
    int width = d->in_dim;
    const int slots = d->conv_type == 0 ? 2 : d->conv_type == 1 ? 4 : d->conv_type == 2 ? 3 : 6;
    const float *const *p = params;
    for (int l = 0; l < d->num_layers; l++) {
        int fin, fout;
        if (d->num_layers == 1) { fin = d->in_dim; fout = d->out_dim; }
        else if (l == 0) { fin = d->in_dim; fout = d->hidden_dim; }
        else if (l == d->num_layers - 1) { fin = d->hidden_dim; fout = d->out_dim; }
        else { fin = d->hidden_dim; fout = d->hidden_dim; }
        int rc = -1;
        switch (d->conv_type) {
        case 0: rc = gnnb_ref_gcn_conv(n, e, cur, nxt, coo, offsets, nbrs, in_deg, out_deg, p[0], p[1], fin, fout); break;
        case 1: rc = gnnb_ref_gin_conv(n, e, cur, nxt, coo, offsets, nbrs, in_deg, out_deg, p[0], p[1], p[2], p[3], d->gin_eps, fin, fout); break;
        case 2: rc = gnnb_ref_sage_conv(n, e, cur, nxt, coo, offsets, nbrs, in_deg, out_deg, p[0], p[1], p[2], fin, fout); break;
        case 3: rc = gnnb_ref_pna_conv(n, e, cur, nxt, coo, offsets, nbrs, in_deg, out_deg, p[0], p[1], p[2], p[3], p[4], p[5], d->pna_delta, fin, fout); break;
        }
        if (rc != 0)
            return -1;
        p += slots;
        if (d->skip && l != 0 && l != d->num_layers - 1)
            for (long i = 0; i < (long)n * fout; i++)
                nxt[i] = nxt[i] + cur[i];
        gnnb_ref_activation(nxt, (long)n * fout, d->activation);
        float *t = cur; cur = nxt; nxt = t;
        width = fout;
    }
    for (int k = 0; k < d->num_pools; k++)
        if (gnnb_ref_global_pool(cur, n, width, d->pools[k], pooled + (size_t)k * width) != 0)
            return -1;
    int din = d->num_pools * width;
    if (din > 1024)
        return -1;
    memcpy(h0, pooled, sizeof(float) * din);
    float *a = h0, *b = h1;
    for (int l = 0; l < d->mlp_num_linear; l++) {
        const int last = (l == d->mlp_num_linear - 1);
        const int dout = last ? d->mlp_out : d->mlp_hidden;
        if (gnnb_ref_linear(a, b, p[0], p[1], din, dout) != 0)
            return -1;
        p += 2;
        if (!last)
            gnnb_ref_activation(b, dout, d->mlp_activation);
        float *t = a; a = b; b = t;
        din = dout;
    }
    memcpy(out, a, sizeof(float) * d->mlp_out);
    return 0;
}

// The reference's per-graph loop (templates/model_tb.cpp.jinja:189-205) over a packed batch.
int gnnb_ref_forward_batched(const ref_desc *d, const float *const *params, const float *x,
                             const int32_t *coo, const int32_t *node_ptr, const int32_t *edge_ptr,
                             int num_graphs, float *out)
{
    static int32_t local[REF_MAX_EDGES * 2];
    for (int g = 0; g < num_graphs; g++) {
        const int n0 = node_ptr[g], n = node_ptr[g + 1] - n0;
        const int e0 = edge_ptr[g], e = edge_ptr[g + 1] - e0;
        if (e > REF_MAX_EDGES)
            return -1;
        for (int i = 0; i < e; i++) {
            local[2 * i] = coo[2 * (size_t)(e0 + i)] - n0;
            local[2 * i + 1] = coo[2 * (size_t)(e0 + i) + 1] - n0;
        }
        int rc = gnnb_ref_forward(d, params, x + (size_t)n0 * d->in_dim, local, n, e,
                                  out + (size_t)g * d->mlp_out);
        if (rc != 0)
            return rc;
    }
    return 0;
}

} // extern "C"
