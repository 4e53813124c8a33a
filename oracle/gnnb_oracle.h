/*
 * gnnb_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's message-passing hot path
 * (sharc-lab/gnn-builder): COO -> degree/neighbour tables -> GCN/GIN/SAGE/PNA
 * conv -> skip/activation -> global pooling -> MLP head.  Single-threaded,
 * scalar, fp32 arithmetic in the reference's loop order.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (libgnnb_hip.so)
 * never links, loads or calls it.
 *
 * Parity status: PINNED.  Every conv is checked by tests/test_oracle_golden.py
 * against the PyG-generated golden vectors the reference commits under
 * gnnbuilder/gnn_builder_lib_test/tb_data (copied as data into
 * tests/golden/ref_tb_data) and against the reference's own C++ kernel library
 * compiled in place (oracle/_ref, see oracle/Makefile).
 *
 * Reference anchors are cited per function as file:line relative to
 * /root/reference/.
 */
#ifndef GNNB_ORACLE_H
#define GNNB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GNNB_O_CONV_GCN = 0, GNNB_O_CONV_GIN = 1, GNNB_O_CONV_SAGE = 2, GNNB_O_CONV_PNA = 3 };
enum { GNNB_O_ACT_RELU = 0, GNNB_O_ACT_GELU = 1, GNNB_O_ACT_SIGMOID = 2, GNNB_O_ACT_TANH = 3, GNNB_O_ACT_NONE = 4 };
enum { GNNB_O_POOL_ADD = 0, GNNB_O_POOL_MEAN = 1, GNNB_O_POOL_MAX = 2 };
/* PNA std flavour: PyG (production parity target) or the HLS library formula. */
enum { GNNB_O_STD_PYG = 0, GNNB_O_STD_HLS = 1 };

/* GCN and explicit self-loop edges (v, v) in the input.  PYG (production parity target): PyG's gcn_norm calls
 * add_remaining_self_loops, which REPLACES the input's self loops by exactly one per node (weight 1), so an
 * explicit self loop neither adds a message nor raises the degree.  HLS: the reference C++ counts the edge in
 * the in-degree and sums its message IN ADDITION to its own self term (gnn_builder_lib.h:1234-1278).  The two
 * agree on inputs without self loops (every molecule set). */
enum { GNNB_O_SELF_LOOPS_PYG = 0, GNNB_O_SELF_LOOPS_HLS = 1 };

/* Same field order as include/gnnb_hip.h's gnnb_model_desc plus pna_std_mode, gcn_self_loop_mode. */
typedef struct gnnb_oracle_desc {
    int32_t conv_type;
    int32_t num_layers;      /* gnn_num_layers (models.py:486) */
    int32_t in_dim;          /* graph_input_feature_dim */
    int32_t hidden_dim;      /* gnn_hidden_dim */
    int32_t out_dim;         /* gnn_output_dim */
    int32_t activation;      /* gnn_activation */
    int32_t skip;            /* gnn_skip_connection */
    int32_t num_pools;
    int32_t pools[3];        /* GlobalPooling.aggrs, order preserved */
    int32_t mlp_num_linear;  /* hidden_layers + 1 */
    int32_t mlp_hidden;
    int32_t mlp_out;
    int32_t mlp_activation;
    float gin_eps;
    float pna_delta;
    int32_t pna_std_mode;
    int32_t gcn_self_loop_mode;
    int32_t output_activation; /* 0 none, 1 softmax, 2 log_softmax over the output vector (models.py:500-502, 572-573) */
    int32_t fpx_w, fpx_i;      /* ap_fixed<W, I> emulation at layer boundaries (code_gen.py:39-52; 0 = float) */
} gnnb_oracle_desc;

/* graph prep: gnn_builder_lib.h:1051-1083, :1086-1124 */
void gnnb_oracle_degree_tables(const int32_t *coo, int n, int e, int32_t *in_deg, int32_t *out_deg);
void gnnb_oracle_neighbor_tables(const int32_t *coo, const int32_t *in_deg, int n, int e,
                                 int32_t *offsets, int32_t *neighbors);

/* y = W x + b for one vector, W [out][in] (gnn_builder_lib.h:808-905) */
void gnnb_oracle_linear(const float *x, float *y, const float *W, const float *b, int in, int out);
void gnnb_oracle_activation(float *x, int64_t count, int kind);

/* single-layer convs on ONE graph; tables from the two prep functions above */
void gnnb_oracle_gcn_conv(int n, const float *x, float *out, const int32_t *offsets,
                          const int32_t *neighbors, const int32_t *in_deg, const float *W,
                          const float *b, int fin, int fout);
void gnnb_oracle_gin_conv(int n, const float *x, float *out, const int32_t *offsets,
                          const int32_t *neighbors, const int32_t *in_deg, const float *W0,
                          const float *b0, const float *W1, const float *b1, float eps, int fin,
                          int hidden, int fout);
void gnnb_oracle_sage_conv(int n, const float *x, float *out, const int32_t *offsets,
                           const int32_t *neighbors, const int32_t *in_deg, const float *Wl,
                           const float *bl, const float *Wr, int fin, int fout);
void gnnb_oracle_pna_conv(int n, const float *x, float *out, const int32_t *offsets,
                          const int32_t *neighbors, const int32_t *in_deg, const float *Wpre,
                          const float *bpre, const float *Wpost, const float *bpost,
                          const float *Wlin, const float *blin, float delta, int std_mode, int fin,
                          int fout);
/* GINE (gnn_builder_lib.h:1126-1166 edge-index table, :1555-1742 conv) */
void gnnb_oracle_neighbor_edge_tables(const int32_t *coo, const int32_t *in_deg, int n, int e, int32_t *offsets,
                                      int32_t *neighbors, int32_t *edge_index);
void gnnb_oracle_gine_conv(int n, const float *x, const float *edge_attr, float *out, const int32_t *offsets,
                           const int32_t *neighbors, const int32_t *edge_index, const int32_t *in_deg,
                           const float *We, const float *be, const float *W0, const float *b0, const float *W1,
                           const float *b1, float eps, int fin, int edge_dim, int hidden, int fout);
/* weight-free convs with committed goldens (gnn_builder_lib.h:2350-2634) */
void gnnb_oracle_simple_conv(int n, const float *x, float *out, const int32_t *offsets,
                             const int32_t *neighbors, const int32_t *in_deg, int f);
void gnnb_oracle_lg_conv(int n, const float *x, float *out, const int32_t *offsets,
                         const int32_t *neighbors, const int32_t *in_deg, int f);

/* out[d] = reduce over the n rows of x[n][d] (gnn_builder_lib.h:2709-2803) */
void gnnb_oracle_global_pool(const float *x, int n, int d, int kind, float *out);

/* number of weight pointers the model consumes, in canonical order:
 * per conv layer: GCN {W,b} | GIN {W0,b0,W1,b1} | SAGE {Wl,bl,Wr} |
 * PNA {Wpre,bpre,Wpost,bpost,Wlin,blin}; then per head linear {W,b}. */
int gnnb_oracle_num_params(const gnnb_oracle_desc *d);

/* whole model on ONE graph with graph-local ids (models.py:551-575) */
int gnnb_oracle_forward(const gnnb_oracle_desc *d, const float *const *params, const float *x,
                        const int32_t *coo, int n, int e, float *out);

/* batched = independent per-graph forwards (SURVEY finding 3).  coo holds
 * batch-global node ids; edges of graph g are rows edge_ptr[g]..edge_ptr[g+1]. */
int gnnb_oracle_forward_batched(const gnnb_oracle_desc *d, const float *const *params,
                                const float *x, const int32_t *coo, const int32_t *node_ptr,
                                const int32_t *edge_ptr, int num_graphs, float *out);

#ifdef __cplusplus
}
#endif
#endif
