/*
 * gnnb_hip.h -- C ABI of the MI355X (gfx950) GNNBuilder runtime, libgnnb_hip.so.
 *
 * This is the drop-in boundary for the reference's one data-parallel hot path:
 * what `gnnbuilder.code_gen.Project` used to emit as Vitis-HLS C++ behind
 *     extern "C" void <name>_top(x, edge_list, out, n, e, copy_params, W...)
 * (gnnbuilder/templates/model.h.jinja:67-79, model.cpp.jinja:686-766) is served
 * here by hand-written HIP kernels behind plain-C entry points: plain pointers
 * and sizes, no torch / C++ types.  `Project.gen_hw_model()` of this package emits
 * a thin `<name>_top` shim over these calls.
 *
 * Semantics = the PyTorch forward of gnnbuilder.models.GNNModel
 * (gnnbuilder/models.py:551-575), applied independently to every graph of a batch.
 * All arithmetic fp32, all indices int32.
 *
 * Conventions
 *   - "_dev" pointers are device (HBM) pointers on the current HIP device;
 *     everything else is host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 *     are asynchronous on that stream unless stated otherwise.
 *   - Every function returns GNNB_OK or a negative gnnb_status; the text of the
 *     last failure on the calling thread is available from gnnb_last_error().
 *   - A model handle is immutable after creation and may be shared by workspaces;
 *     a workspace serves one in-flight call at a time (the reference's statics made
 *     the whole kernel non-re-entrant: model.cpp.jinja:5-22).
 */
#ifndef GNNB_HIP_H
#define GNNB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNB_VERSION 104

typedef enum gnnb_status {
    GNNB_OK = 0,
    GNNB_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    GNNB_ERR_CAPACITY = -2,  /* batch exceeds the workspace (reference: silent overflow) */
    GNNB_ERR_HIP = -3,       /* a HIP runtime call failed */
    GNNB_ERR_NO_DEVICE = -4, /* no gfx950 device visible */
    GNNB_ERR_GRAPH = -5,     /* malformed batch: an edge leaves its graph, ptr not monotone, a broken max_graph_nodes
                              * promise, a large-segment triple that disagrees with the ptr arrays */
    GNNB_ERR_RANGE = -6      /* a REDUCED-precision math mode (gnnb_model_desc::math 2 / 3) produced a non-finite value: fp16's
                              * range (65504) was exceeded by an activation or a weight, or the inputs were not finite.  The
                              * results of that forward are unspecified; run the model with math = 0 (gnnb_workspace_check) */
} gnnb_status;

/* gnnbuilder/models.py:453-459 (SUPPORTED_GNN_CONVS; GAT has no native path in the
 * reference either: gnn_builder_lib.h:2343) */
typedef enum gnnb_conv { GNNB_CONV_GCN = 0, GNNB_CONV_GIN = 1, GNNB_CONV_SAGE = 2, GNNB_CONV_PNA = 3 } gnnb_conv;
/* gnnbuilder/models.py:362 (SUPPORTED_ACTIVATIONS); GELU is the exact erf form */
typedef enum gnnb_act { GNNB_ACT_RELU = 0, GNNB_ACT_GELU = 1, GNNB_ACT_SIGMOID = 2, GNNB_ACT_TANH = 3, GNNB_ACT_NONE = 4 } gnnb_act;
/* gnnbuilder/models.py:317-321 (SUPPORTED_GLOBAL_POOLING_AGGRS) */
typedef enum gnnb_pool { GNNB_POOL_ADD = 0, GNNB_POOL_MEAN = 1, GNNB_POOL_MAX = 2 } gnnb_pool;

/* GNNModel.output_activation (gnnbuilder/models.py:500-502, 572-573): a module built with dim=-1, i.e. a softmax-like
 * map over the model's output vector */
typedef enum gnnb_out_act { GNNB_OUT_NONE = 0, GNNB_OUT_SOFTMAX = 1, GNNB_OUT_LOG_SOFTMAX = 2 } gnnb_out_act;

/* What GNNModel.__init__ fixes (gnnbuilder/models.py:463-549). */
typedef struct gnnb_model_desc {
    int32_t conv_type;      /* gnnb_conv */
    int32_t num_layers;     /* gnn_num_layers, 0..GNNB_MAX_LAYERS */
    int32_t in_dim;         /* graph_input_feature_dim */
    int32_t hidden_dim;     /* gnn_hidden_dim */
    int32_t out_dim;        /* gnn_output_dim */
    int32_t activation;     /* gnn_activation (gnnb_act), applied after EVERY conv */
    int32_t skip;           /* gnn_skip_connection: middle layers only (models.py:562-564) */
    int32_t num_pools;      /* 1..3 */
    int32_t pools[3];       /* GlobalPooling.aggrs in order (gnnb_pool) */
    int32_t mlp_num_linear; /* MLP.hidden_layers + 1 */
    int32_t mlp_hidden;     /* MLP.hidden_dim */
    int32_t mlp_out;        /* MLP.out_dim = model output width */
    int32_t mlp_activation; /* MLP.activation (gnnb_act), between head linears only */
    float gin_eps;          /* GINConv_GNNB.eps (models.py:76) */
    float pna_delta;        /* PNAConv_GNNB.delta used verbatim as avg_deg_log (models.py:236) */
    int32_t output_activation; /* gnnb_out_act, applied to every graph's output row */
    /* Fixed-point emulation of the reference's float_or_fixed = "fixed" build (code_gen.py:39-52 FPX(W, I);
     * model.h.jinja:38-62: F_TYPE = W_TYPE = ap_fixed<W, I>, quantisation AP_TRN, overflow AP_WRAP).  fpx_w = 0: float
     * (default).  Otherwise the input features, every weight and bias, every conv layer's output (after skip and
     * activation), the pooled vector and every head layer's output are put on the ap_fixed<W, I> grid:
     *     q(v) = wrap(floor(v * 2^(W-I)) / 2^(W-I))   into [-2^(I-1), 2^(I-1)).
     * Sums inside a layer are carried in fp32 (the HLS build rounds every partial result to W bits -- `linear` accumulates
     * in F_TYPE per multiply-add, gnn_builder_lib.h:866-904): this is a LAYER-BOUNDARY study tool for accuracy against
     * width, NOT a bit-exact ap_fixed model, and it is pinned to nothing of the reference (ap_fixed.h is a Vitis header:
     * the fixed-point build of the library does not compile here).  What IS guaranteed, and tested: outputs lie on the
     * grid; the HIP path and the C oracle's emulation (which differ only in the fp32 summation order inside a layer) are
     * never more than two grid steps apart, with >= 95 % of the outputs identical for W <= 16; at W - I = 20 the finest
     * format reproduces the float model to 1e-3.  The fused stack kernels are not used in this mode. */
    int32_t fpx_w;
    int32_t fpx_i;
    /* Arithmetic of the dense updates -- a property of the DESIGN, as float_or_fixed / fpx are in the reference
     * (code_gen.py:63-82 Project(float_or_fixed, fpx) baked into model.h.jinja:38-62), captured by gnnb_model_create
     * and used by every launch of this model's forwards and of its workspaces' graph preps, whatever other models of
     * the process use and whatever gnnb_set_option("math", ...) is called later, from any thread:
     *    0  native fp32 MFMA everywhere (what bench.py's `value` runs)
     *    1  "bf16x6": fp32-equivalent, six bf16 MFMA products of an exact 3-way split
     *    2  "bf16x3": REDUCED precision (~18 significant bits per product)
     *    3  "f16x3":  REDUCED precision (~22 bits) with fp16's RANGE -- see "math" under gnnb_set_option below
     *   -1  not a property of this model: every launch follows the process-wide option "math" as it stands AT THAT
     *       LAUNCH (the pre-103 behaviour; what the Python runtime passes unless told otherwise, for A/B measurements)
     * (version 103; a zero-initialised description is a native-fp32 design.)  In modes 2 and 3 every reduced kernel
     * checks what it produces: a non-finite value sets flag 64 of the workspace, reported as GNNB_ERR_RANGE by
     * gnnb_workspace_check (and lazily by the next gnnb_graph_prep on that workspace). */
    int32_t math;
} gnnb_model_desc;

#define GNNB_MAX_LAYERS 16

typedef struct gnnb_model gnnb_model;
typedef struct gnnb_workspace gnnb_workspace;

/* ------------------------------------------------------------------ library / device */
int gnnb_version(void);
const char *gnnb_last_error(void);
/* number of gfx950 devices visible; <= 0 means the product path cannot run */
int gnnb_device_count(void);
/* synchronise a stream (hipStreamSynchronize) */
int gnnb_stream_sync(void *stream);

/* ------------------------------------------------------------------ model
 * Number of weight tensors the description consumes, in canonical order:
 *   per conv layer  GCN  {W[out,in], b[out]}
 *                   GIN  {W0[out,in], b0[out], W1[out,out], b1[out]}        (hidden = out, models.py:90)
 *                   SAGE {Wl[out,in], bl[out], Wr[out,in]}
 *                   PNA  {Wpre[in,2in], bpre[in], Wpost[out,13in], bpost[out], Wlin[out,out], blin[out]}
 *   then per head linear {W[out,in], b[out]}.
 * All row-major [out][in] fp32 = torch.nn.Linear.weight (gnn_builder_lib.h:40-45). */
int gnnb_model_num_params(const gnnb_model_desc *desc);
/* Copies the weights to the device once (the reference's copy_parameters_flag=1 call,
 * model.cpp.jinja:724-730).  host_params[i] points at tensor i (host memory). */
int gnnb_model_create(const gnnb_model_desc *desc, const float *const *host_params, int num_params,
                      gnnb_model **out_model);
void gnnb_model_destroy(gnnb_model *model);
int gnnb_model_get_desc(const gnnb_model *model, gnnb_model_desc *out_desc);

/* ------------------------------------------------------------------ workspace
 * Device scratch for batches of at most max_graphs graphs / max_nodes nodes /
 * max_edges edges in total (replaces MAX_NODES/MAX_EDGES static arrays,
 * model.cpp.jinja:5-22, but checked: GNNB_ERR_CAPACITY instead of overflow). */
int gnnb_workspace_create(const gnnb_model *model, int max_graphs, int max_nodes, int max_edges,
                          gnnb_workspace **out_ws);
void gnnb_workspace_destroy(gnnb_workspace *ws);
size_t gnnb_workspace_bytes(const gnnb_workspace *ws);
/* Promise that no graph of the batches run on this workspace has more than `n` nodes (0 = no
 * promise, the default).  Small molecules (n <= 61; 45 in the opt-in bf16x6 math mode; for a 2-layer fp32 GCN 169 with in_dim <= 16 and 89 with in_dim 17..32) let whole graphs be staged in LDS, which enables
 * the fused conv-stack kernels.  The promise is VALIDATED on the device by every graph prep: a
 * larger graph makes gnnb_workspace_check() return GNNB_ERR_GRAPH (the reference's MAX_NODES, by
 * contrast, is never checked: model.cpp.jinja:5-22).  Callers that never call the check still find out: the NEXT
 * gnnb_graph_prep / gnnb_forward_batched on the workspace after a flagged batch has run returns GNNB_ERR_GRAPH
 * (read from a host-mapped word, no synchronisation; best effort -- the check is the authoritative answer). */
int gnnb_workspace_set_max_graph_nodes(gnnb_workspace *ws, int n);
/* Promise that no node of the batches run on this workspace has an in-degree above `d` (0 = no promise, the default).  The
 * reference knows a per-design degree figure only as a hint (Project(..., degree_guess, ...), code_gen.py:63-82: HLS loop trip
 * counts); here it is a BOUND, checked.  PNA's scalers
 * depend on the in-degree only (amp = log(d + 1) / delta, att = its reciprocal, gnn_builder_lib.h:1857-1875), so under a
 * promise of d <= 15 (molecules: <= 6) a PNA layer's 13 F-wide post-NN product [x | A | amp A | att A] . W^T is evaluated as
 * the 5 F-wide [x | A] . (W_x | W_1 + amp(d) W_2 + att(d) W_3)^T with the rows sorted by degree class and one pre-combined
 * weight matrix per class (formed in double at gnnb_model_create): the same mathematics, 2.6x fewer flops.  Applies when
 * the batch is prepared with the model's own pna_delta.  VALIDATED on the device by every graph prep like the node
 * promise (flag 32 of gnnb_workspace_check).  Other conv types ignore it. */
int gnnb_workspace_set_max_degree(gnnb_workspace *ws, int d);

/* Which kernels the LAST forward on this workspace ran (diagnostics / benchmarks: the answer does not change any
 * result).  The reference has one dataflow per generated model (compute_gnn_head, model.cpp.jinja:151-359); here the
 * same model can take the LDS-resident stack kernels or the layer-by-layer kernels depending on the batch (the
 * max_graph_nodes promise) and the options. */
enum {
    GNNB_PATH_NONE = 0,      /* no forward yet */
    GNNB_PATH_LAYERWISE = 1, /* per layer: gather-aggregate + GEMM kernels */
    GNNB_PATH_STACK = 2,     /* whole conv stack + pooling in k_gcn2_fused (GCN / GIN, graphs <= 61 nodes) */
    GNNB_PATH_STACK_ZF = 3,  /* 2-layer fp32 GCN in k_gcn2_zf (last layer transformed before aggregation, graphs <= 169 nodes for in_dim <= 16, <= 89 for in_dim 17..32) */
    GNNB_PATH_LARGE_LAYERWISE = 16 /* flag, or-ed to a STACK value: the batch had a large segment that ran layer by layer */
};
int gnnb_workspace_last_path(const gnnb_workspace *ws);

/* Large segment.  The stack kernels stage WHOLE graphs in LDS, hence the max_graph_nodes promise.  A batch that holds a few
 * graphs beyond it (real ogbg-molhiv: 99 % of the molecules have <= 57 atoms, the largest 222) need not give the path up:
 * the caller orders the batch so that those graphs come LAST and names the first of them -- graph index, its first node
 * row and its first edge row (host integers: launch sizes depend on them).  Graphs [0, first_graph) keep the promise
 * (still validated on the device) and run in the stack kernel; graphs [first_graph, num_graphs) are unrestricted and run
 * through the layer-by-layer kernels; both halves fill one pooled matrix and share the readout.  The setting applies to
 * every following gnnb_graph_prep / forward on the workspace until it is changed; first_graph < 0 removes it.  The
 * triple must describe the batch it is used with: every graph prep checks ON THE DEVICE that first_node ==
 * node_ptr[first_graph] and first_edge == edge_ptr[first_graph] and flags the batch otherwise (GNNB_ERR_GRAPH from
 * gnnb_workspace_check, flag bit 16) -- a triple left over from the batch before would otherwise leave the rows between the
 * two boundaries to neither half.  The
 * reference has no counterpart: its MAX_NODES is a compile-time array bound of the generated model
 * (templates/model.cpp.jinja:5-22), any graph up to it takes the same dataflow.
 * (gnnbuilder_amd.batching.order_large_last does the ordering and returns the three numbers and the permutation.) */
int gnnb_workspace_set_large_segment(gnnb_workspace *ws, int first_graph, int first_node, int first_edge);

/* ------------------------------------------------------------------ batched forward
 * x_dev        [num_nodes, in_dim] fp32 row-major, graphs concatenated
 * coo_dev      [num_edges, 2] int32 (src, dst) with BATCH-GLOBAL node ids (edge_index.T as
 *              code_gen.py:262 writes it, plus the graph's node offset); message flows src->dst
 * node_ptr_dev [num_graphs+1] int32, node range of graph g = [node_ptr[g], node_ptr[g+1])
 * edge_ptr_dev [num_graphs+1] int32, edge rows of graph g (edges are grouped by graph,
 *              original order kept inside a graph)
 * out_dev      [num_graphs, mlp_out] fp32
 * Runs graph prep (degree / CSR-by-destination tables, gnn_builder_lib.h:1051-1124), the
 * conv stack, pooling and the MLP head on `stream`. */
int gnnb_forward_batched(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev,
                         const int32_t *coo_dev, const int32_t *node_ptr_dev,
                         const int32_t *edge_ptr_dev, int num_graphs, int num_nodes, int num_edges,
                         float *out_dev, void *stream);
/* Same with the tables of a previous gnnb_graph_prep() on this workspace re-used
 * (the topology of the batch is unchanged; only features differ). */
int gnnb_forward_prepared(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev,
                          float *out_dev, void *stream);
/* Software-pipelined form of gnnb_forward_batched for a stream of batches over TWO workspaces of one model (ABI 104):
 * the forward of the batch prepared on `ws` (as gnnb_forward_prepared), then the graph prep of the NEXT batch on `ws_next`
 * (as gnnb_graph_prep with the model's pna_delta), both on `stream` -- one call per batch,
 *     gnnb_graph_prep(wsA, batch 0);  prep_next(wsA, x0, out0, wsB, batch 1);  prep_next(wsB, x1, out1, wsA, batch 2); ...
 *     ... gnnb_forward_prepared(ws?, x_last, out_last)
 * Results and tables are bit-identical to gnnb_forward_batched per batch.  What it buys: where ws_next carries a
 * max_graph_nodes promise <= 64, no large segment, and nothing has to be launched behind its tables (no degree classes, stage
 * cuts or GCN coefficient table), the next batch's prep runs as EXTRA WORKGROUPS of the forward's readout kernel (k_head_small)
 * instead of as a launch of its own: -6.6 % per forward with one batch in flight at BASELINE config 2; with three batches in
 * flight on three streams no gain, so bench.py's `value` keeps gnnb_forward_batched.  Anywhere else the prep is
 * launched behind the forward, as the two calls would.  The reference has no counterpart (its compute_degree_tables /
 * compute_neighbor_tables run serially in front of every graph's layers, gnn_builder_lib.h:1051-1124); option guest_prep = 0
 * turns the in-kernel form off.
 * ws_next must differ from ws (GNNB_ERR_INVALID) and, like any workspace, must not be in use by work still running on
 * another stream.  Errors of the next batch's host-side validation (capacity, a flag left by an earlier batch on ws_next) are
 * returned before anything is enqueued; after a forward error ws_next has no prepared batch. */
int gnnb_forward_prepared_prep_next(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev, float *out_dev,
                                    gnnb_workspace *ws_next, const int32_t *coo_dev, const int32_t *node_ptr_dev,
                                    const int32_t *edge_ptr_dev, int num_graphs, int num_nodes, int num_edges, void *stream);
/* Host-buffer convenience: H2D copies, forward, D2H copy, synchronises. */
int gnnb_forward_batched_host(const gnnb_model *model, gnnb_workspace *ws, const float *x,
                              const int32_t *coo, const int32_t *node_ptr, const int32_t *edge_ptr,
                              int num_graphs, int num_nodes, int num_edges, float *out);
/* Device-side validation result of the last prep on this workspace (synchronises the
 * stream): GNNB_OK or GNNB_ERR_GRAPH. */
int gnnb_workspace_check(gnnb_workspace *ws, void *stream);

/* ------------------------------------------------------------------ stage entry points
 * The individual kernels, for parity tests, profiling and the roofline measurement. */

/* compute_degree_tables + compute_neighbor_tables for a whole batch
 * (gnn_builder_lib.h:1051-1124): fills the workspace's row_ptr[N+1], col[E]
 * (CSR by destination, stable in COO order), in-degree, node tiles.
 *
 * Explicit self loops.  The tables belong to the MODEL the workspace was created for: on a GCN workspace an edge (v, v) of
 * the input is NOT entered (PyG's gcn_norm replaces the self loops of the input by exactly one per node, which every GCN
 * aggregate adds itself; the reference C++ would count it on top of its own self term, gnn_builder_lib.h:1234-1278, and
 * disagrees with its own PyTorch golden there).  Consequences for the stage entry points: on a GCN workspace
 * gnnb_aggregate(GNNB_AGG_SUM / MEAN / LG / SIMPLE) and gnnb_aggregate_edges do not see those edges either; on any other
 * workspace gnnb_aggregate(GNNB_AGG_GCN) counts an explicit self loop as an ordinary edge.  Dropped edges leave a gap at
 * the end of their graph's CSR segment: row_ptr holds row STARTS (a row's length is in_deg, not a difference of row_ptr),
 * and the unused col / edge-index slots of a segment read -1 in the *_to_host copies.
 *
 * A malformed batch (GNNB_ERR_GRAPH from gnnb_workspace_check) is reported ONCE: the check, or the lazy report of the next
 * gnnb_graph_prep on the workspace -- which returns GNNB_ERR_GRAPH WITHOUT having enqueued the batch it was called with --
 * clears the device flag, so a later check speaks about later batches only. */
int gnnb_graph_prep(gnnb_workspace *ws, const int32_t *coo_dev, const int32_t *node_ptr_dev,
                    const int32_t *edge_ptr_dev, int num_graphs, int num_nodes, int num_edges,
                    float pna_delta, void *stream);
/* copy the tables of the last prep to host arrays (any may be NULL); synchronises */
int gnnb_graph_tables_to_host(gnnb_workspace *ws, int32_t *row_ptr /*[N+1]*/, int32_t *col /*[E]*/,
                              int32_t *in_deg /*[N]*/, void *stream);

typedef enum gnnb_agg {
    GNNB_AGG_GCN = 0,  /* sum_j x_j/sqrt((1+d_i)(1+d_j)) + x_i/(1+d_i)   gnn_builder_lib.h:1213-1289 */
    GNNB_AGG_SUM = 1,  /* sum_j x_j + (1+eps) x_i                         gnn_builder_lib.h:1389-1437,1525-1535 */
    GNNB_AGG_MEAN = 2, /* mean_j x_j (0 if no neighbour)                  gnn_builder_lib.h:2161-2209 */
    GNNB_AGG_PNA = 3,  /* [max|min|mean|std]_j (q_i + p_j), out width 4w  gnn_builder_lib.h:1750-1834, PyG std:
                        * sqrt(max(E[h^2] - E[h]^2, 1e-5)), 0 where <= sqrt(1e-5) -- so a node of in-degree 0 or 1 gets std 0
                        * (and max = min = mean = 0 at in-degree 0).  The reference holds no golden vector for such nodes
                        * (its fixture graph has in-degrees 2..11); the values follow from the formula that reproduces
                        * tb_pna_output.bin to 1.2e-7 */
    GNNB_AGG_LG = 4,   /* sum_j x_j/sqrt(d_i d_j), no self term, 0 where d_i d_j = 0   gnn_builder_lib.h:2350-2499 (lg_conv) */
    GNNB_AGG_SIMPLE = 5, /* sum_j x_j, no self term                           gnn_builder_lib.h:2501-2634 (simple_conv) */
    GNNB_AGG_COPY = 6  /* out = x: the kernel's launch shape and bytes with no gather (calibration of the roofline) */
} gnnb_agg;
/* Gather-aggregate over the prepared batch.  x_dev [N,width]; out_dev [N,width]
 * ([N,4*width] for PNA).  self_dev: PNA only, the per-destination term q [N,width] (NULL: none -- the statistics of p_j alone,
 * what the degree-class form of gnnb_workspace_set_max_degree aggregates)
 * (NULL otherwise).  eps: GIN's epsilon (SUM only). */
int gnnb_aggregate(gnnb_workspace *ws, int agg_kind, const float *x_dev, const float *self_dev,
                   float *out_dev, int width, float eps, void *stream);

/* PNA's source-half pre-NN product and its aggregate in one kernel (round 5): out [N, 4*width] = max | min | mean | std over
 * every node's sources j of p_j = Wb x_j, Wb [width, ldw] row-major = the x_j half of pre_nns.0.0.weight (columns width ..
 * 2 width - 1: pass weight + width, ldw = 2 width) -- the reference's per-edge `linear` (gnn_builder_lib.h:1807) split per
 * node, + pna_conv_agg (:1750-1834) without a destination term (the degree-class form folds that term into the post-NN's
 * weights).  p never goes to HBM: whole graphs are staged in LDS, so the workspace needs a max_graph_nodes promise that fits
 * a 64-row stage; widths 128 / 64 / 32.  The forward takes this route by itself (option pna_pagg); this entry exists for
 * measurements and tests. */
int gnnb_pna_product_aggregate(gnnb_workspace *ws, const float *x_dev, const float *wb_dev, int ldw, float *out_dev, int width,
                               void *stream);

/* GINE aggregate (gine_conv_agg + the self term of gine_conv, gnn_builder_lib.h:1555-1742):
 *   out_i = (1 + eps) x_i + sum_{j->i} relu(x_j + edge_term[e]),  e = the COO row of the edge j->i.
 * edge_term_dev [E, width] holds the projected edge features W_e e + b_e in COO (input) order -- produce it with
 * gnnb_linear over the [E, edge_dim] edge-feature matrix.  The CSR slot -> COO row map is the reference's
 * edge_index_table (compute_neighbor_and_edge_index_tables, gnn_builder_lib.h:1126-1166), written by graph prep. */
int gnnb_aggregate_edges(gnnb_workspace *ws, const float *x_dev, const float *edge_term_dev, float *out_dev,
                         int width, float eps, void *stream);
/* copy that table of the last prep to the host ([E] int32, batch-global COO rows); synchronises */
int gnnb_edge_index_table_to_host(gnnb_workspace *ws, int32_t *edge_index_table /*[E]*/, void *stream);

/* Dense update on the matrix cores: for up to 4 K-segments s,
 *   Y[M,N] = act( sum_s (rowscale_s[m] * A_s[M,K_s]) . W[:, koff_s : koff_s+K_s]^T + bias + skip )
 * W is row-major [N, ldw] (torch Linear layout); A_s row-major with leading dim lda_s.
 * One segment with rowscale NULL is a plain batched `linear` (gnn_builder_lib.h:808-905). */
typedef struct gnnb_gemm_seg {
    const float *a_dev;        /* [M, lda] */
    const float *rowscale_dev; /* [M] or NULL */
    int32_t lda;
    int32_t k;                 /* K_s */
} gnnb_gemm_seg;
int gnnb_linear(const gnnb_gemm_seg *segs, int num_segs, const float *w_dev, int ldw,
                const float *bias_dev /*[N] or NULL*/, const float *skip_dev /*[M,N] or NULL*/,
                float *y_dev /*[M,N]*/, int M, int N, int act, void *stream);

/* Diagnostics for the large-K GEMM's stream-K tail (K >= 1024: a tile's K range may be shared by several workgroups, which
 * park partial sums in a scratch and count their arrivals per shared tile).  The scratch belongs to the WORKSPACE whose
 * forward launches the GEMM (allocated with it, freed with it: forwards of different workspaces -- on any streams, eager or
 * replayed from hipGraphs -- never share one); the standalone gnnb_linear has one per (device, stream) and never uses it
 * while `stream` is being captured (a captured gnnb_linear takes row slices).  This call reads the scratch's arrival
 * counters and the guard region behind them back: GNNB_OK when every counter is zero (as each launch must leave them) and
 * the guard is untouched, GNNB_ERR_INVALID otherwise.  ws == NULL: the standalone scratch of (current device, stream).
 * Synchronises `stream`.  No reference counterpart (gnn_builder_lib.h:808-905 `linear` is one scalar loop per node). */
int gnnb_debug_stream_k_guard(gnnb_workspace *ws, void *stream);

/* global_{add,mean,max}_pool per graph, concatenated in `pools` order
 * (gnn_builder_lib.h:2709-2803, model.cpp.jinja:440-448): x_dev [N,d] -> out_dev [B, num_pools*d] */
int gnnb_global_pool(gnnb_workspace *ws, const float *x_dev, int d, const int32_t *pools,
                     int num_pools, float *out_dev, void *stream);

/* ------------------------------------------------------------------ timing helpers
 * hipEvent wrappers so the C harness and bench.py time kernels on the stream they run on. */
int gnnb_event_create(void **out_event);
int gnnb_event_record(void *event, void *stream);
int gnnb_event_elapsed_ms(void *start, void *stop, float *out_ms); /* synchronises on stop */
void gnnb_event_destroy(void *event);

/* Launch gnnb_aggregate `iters` times back to back from C (host launch cost ~3.6 us, below the
 * kernel's duration, so the stream stays busy) rotating over `nbuf` input/output buffers, and
 * return the mean wall time per launch between two HIP events on `stream`.  Measurement aid for
 * bench.py's roofline object; synchronises. */
int gnnb_aggregate_timed(gnnb_workspace *ws, int agg_kind, const float *const *x_dev_list,
                         const float *self_dev, float *const *out_dev_list, int nbuf, int width,
                         float eps, int iters, void *stream, float *out_us_per_launch);
/* Same for one gnnb_linear configuration (single segment). */
int gnnb_linear_timed(const float *a_dev, int lda, int k, const float *w_dev, int ldw,
                      const float *bias_dev, float *y_dev, int M, int N, int act, int iters,
                      void *stream, float *out_us_per_launch);

/* Same for the fused GCN stack + pooling kernel on the workspace's prepared batch (the kernel
 * gnnb_forward_prepared runs when the model is a GCN or a GIN (hidden 32 / 64 / 128, out <= hidden) of two or more layers and a
 * max_graph_nodes promise is set;
 * replaces compute_gnn_head + compute_global_graph_pooling, templates/model.cpp.jinja:151-359,
 * :413-449).  GNNB_ERR_INVALID when that path is not eligible. */
int gnnb_gcn_stack_timed(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev, int iters,
                         void *stream, float *out_us_per_launch);

/* device memory helpers for hosts without another allocator (the generated C harness) */
int gnnb_malloc(void **out_dev, size_t bytes);
void gnnb_free(void *dev);
int gnnb_memcpy_h2d(void *dst_dev, const void *src, size_t bytes, void *stream);
int gnnb_memcpy_d2h(void *dst, const void *src_dev, size_t bytes, void *stream);

/* tuning knobs (also read from the environment at load as GNNB_<NAME>): they select between parity-tested forms of the
 * same computation and never change results beyond fp32 rounding.
 *   tile_rows (>= 4, default 8)  node-tile size of graph prep         agg_lds_kb, agg_ring_waves / _slots / _wg_per_cu,
 *   agg_nt_store, agg_balance    launch shape of the gather-aggregate  gemm_variant, gemm_dma, gemm_tail_split,
 *   gemm_wlds, gemm_wlds_slots, gemm_max_wg_per_cu  which GEMM kernel   fuse_narrow, fuse_gcn2, fuse_head, head_small,
 *   head_split                   which launches are fused (0 = layer by layer / separate readout)
 *   first_ring (default 1)       a narrow-input first layer (F_in <= 32) in ring form: whole graphs staged in LDS, all N <= 256
 *                                output columns from one stage (k_conv_first); 0 = inside the GEMM's A stage (k_linear_reg)
 *   fuse_pool (default 1)        global pooling in the epilogue of the last conv layer's GEMM where that GEMM has one
 *                                (GraphSAGE's large-K segmented GEMM): its [N, d] output is never written; 0 = pooling pass
 *   agg_balance (default 0)      gather-aggregate workgroups take row-balanced ranges (boundary graphs staged twice) instead
 *                                of whole-graph runs: measured slower (15.9 vs 15.2 us at BASELINE config 2), kept opt-in
 *   fuse_zf (default 1)          2-layer fp32 GCN stacks through k_gcn2_zf (last layer transformed before it is aggregated);
 *                                0 = k_gcn2_fused.  zf_shape: 0 = two 8-wave workgroups per CU, 96-row stages; 1 = one 16-wave
 *                                workgroup, 176-row stages; 2 (default) = 1 wherever it exists (input widths <= 16)
 *   gemm_tail_split (default 2)  the last, partial round of tiles of the large-K GEMM (k_linear_dma): 2 = for K >= 1024 cut
 *                                along K into equal runs over all resident workgroups, parts added up in run order by the last
 *                                workgroup at each tile (stream-K: one summation order per shape, not the unsplit one; its 64 MB
 *                                of scratch belong to the workspace whose forward launches the GEMM -- the stand-alone
 *                                gnnb_linear keeps one per (device, stream) and takes row slices while that stream is being
 *                                captured), row slices otherwise; 1 = row slices (bit-identical to 0); 0 = whole tiles
 *   pna_fold_lin (default 1)     PNA: `lin` folded into the post-NN at upload (W' = W_lin W_post, formed in double): one 13F-wide
 *                                GEMM per layer with skip + activation (+ the last layer's pooling) in its epilogue; 0 = the
 *                                reference's two products.  Off under the fixed-point emulation
 *   pna_classes (default 1)      PNA under a max_degree promise (gnnb_workspace_set_max_degree): the degree-class form; 0 = the
 *                                general 13 F-wide form
 *   pna_pagg (default 1)         PNA under both promises (max_degree: no destination term; max_graph_nodes: whole graphs fit a 64-row
 *                                stage): a full-width layer's source-half product and its aggregate in one kernel, the per-node
 *                                messages never in HBM (k_pna_pagg); 0 = GEMM + aggregate kernel
 *   pna_first (default 1)        PNA with the max_graph_nodes promise: a narrow-input layer (F <= 12: the first) as ONE kernel --
 *                                pre-NN, statistics, scalers, 13F-wide post-NN (k_pna_first); 0 = layer-by-layer kernels
 *   sage_first_mean (default 1)  GraphSAGE with the max_graph_nodes promise (<= 49): the narrow first layer keeps its output rows in
 *                                LDS and forms the next layer's mean aggregate there (k_sage_first_mean); 0 = k_conv_first + the
 *                                aggregate kernel (bit-identical)
 *   stage_cut (default 0)        the deep-GCN / GIN stack kernel's workgroups take whole stages of the batch's global greedy stage list,
 *                                planned by graph prep (k_stage_cut_lds, 15-47 us): the kernel alone 8 % faster, the three-stream
 *                                pipeline 2 % at best and 5 % slower with the planner launch: for one-at-a-time forwards
 *                                (its tables are carved at gnnb_workspace_create: the option must be on THEN; a workspace created
 *                                without them keeps equal tile counts)
 *   zf_head (default 0)          the 2-layer GCN stack kernel also runs the MLP head on the graphs it pooled (one launch for conv
 *                                stack + pooling + head): measured slower than the separate readout
 *   prep_group (default 4)       graph prep, molecule path (max_graph_nodes promise <= 64), batches of >= 2047 graphs: 4 = a wave
 *                                prepares four consecutive graphs with its fetches batched (a quarter of the workgroups: cheaper
 *                                beside the stack kernels of other batches in flight); 1 = one graph per wave (lower latency when
 *                                the batch has the chip to itself).  Same tables
 *   head_pairs (default 1)       the readout on a pooled matrix (k_head_small): 1 = its MFMA operands in pairs, 72 registers -- a readout
 *                                wave and a graph-prep wave of another batch then share the registers the 2-layer GCN stack kernel
 *                                leaves on a SIMD (batches in flight on several streams); 0 = four operand slices in flight, 82
 *                                registers: ~2 us faster when one batch has the chip to itself.  Same bits.  Together with
 *                                prep_group = 1: the settings for ONE stream of forwards
 *   guest_prep (default 1)       gnnb_forward_prepared_prep_next: 1 = the next batch's prep as extra workgroups of the readout
 *                                kernel where eligible; 0 = always a launch of its own behind the forward
 *   agg_form (default 0)         gather-aggregate kernel: 0 = LDS ring, 1 = barrier-free register gather (k_aggregate_rg) wherever
 *                                it exists, 2 = that form for PNA only; agg_rg_r / agg_rg_wgs / agg_rg_flags shape its launch
 *   fold_skip (default 1)        GraphSAGE: a middle layer's skip connection (y = conv(x) + x) as + I on the root weights -- x is an
 *                                operand of the layer's GEMM anyway -- instead of a second read of x in the epilogue (PNA's folded
 *                                forms always carry it); 0 = the skip operand
 *   large_fork (default 2)       how a batch's large segment (gnnb_workspace_set_large_segment) runs: 2 = small per-layer
 *                                kernels behind the stack kernel, 1 = the same on a side stream, 0 = the big layer-wise kernels
 * "math": the PROCESS-WIDE default of the math mode -- what the stand-alone entries (gnnb_linear, ...) run in and what models
 * created with gnnb_model_desc::math = -1 follow at every launch; a model whose description names a mode (0 .. 3) never reads
 * it (version 103: two designs of different precision in one process, or a thread calling gnnb_set_option, do not change each
 * other's arithmetic).  0 (default) = native fp32 MFMA everywhere; 1 = every wide update (the fused GCN stack's A1.W1^T, the
 * K <= 128 GEMMs, the large-K segmented GEMM) as six bf16 MFMA products of an exact 3-way bf16 split of both fp32
 * operands, fp32 accumulate (results at fp32 rounding level; GNNB_MATH=1); stack kernels that are faster
 * in fp32 than any bf16x6 form (k_gcn2_zf, the GIN / deep stacks) keep running: the mode is never slower than 0.
 * 2 = REDUCED precision ("bf16x3", GNNB_MATH=2; the accuracy-vs-throughput analogue of the reference's float_or_fixed switch,
 * code_gen.py:39-52): as 1, and the 2-layer GCN stack kernel k_gcn2_zf (input widths <= 16) multiplies H.W1^T as three bf16
 * MFMA products on round-to-nearest hi + mid bf16 pieces of both operands, fp32 accumulate -- ~18 significant bits per product
 * (outputs within ~3e-6 of their scale of the fp32 form at BASELINE config 2).
 * 3 = as 2 with fp16 pieces ("f16x3", GNNB_MATH=3), also in the LDS-DMA GEMMs and in the GIN / deep-GCN stack kernel (every
 * BASELINE config has a reduced form): ~22 significant bits per product at the same speed or better (7.5e-8 against
 * a float64 evaluation at BASELINE config 2, where the fp32 form has 7.1e-8) -- but fp16's RANGE: hidden activations or
 * weights of 65504 and above turn into inf, and pieces below 6e-8 are lost (an absolute floor per operand element).
 * Unknown names or values out of range return GNNB_ERR_INVALID.  Thread safety (version 103): every knob is one relaxed atomic
 * word -- setting one while other threads launch is no data race; the knobs only choose between kernels that give the same
 * results, except "math", which a model with its own gnnb_model_desc::math never reads. */
int gnnb_set_option(const char *name, int value);

#ifdef __cplusplus
}
#endif
#endif /* GNNB_HIP_H */
