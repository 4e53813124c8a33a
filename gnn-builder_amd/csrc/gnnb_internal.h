// gnnb_internal.h -- launcher prototypes shared by the kernel and runtime translation units
// of libgnnb_hip.so (gfx950 only).  Public ABI: include/gnnb_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "gnnb_hip.h"

namespace gnnb {

// Tables produced by graph prep for one batch (all device pointers, owned by a workspace).
struct BatchTables {
    int32_t *row_ptr;    // [N+1] START of every node's CSR row (batch-global); the row's length is node_rec[2v].y --
                         //       dropped edges leave a gap at the end of a graph's segment, so differences are not degrees
    int32_t *col;        // [E]   source node (batch-global id) of every in-edge, stable COO order
    int32_t *eid;        // [E]   COO row of every CSR slot (the reference's edge_index_table, gnn_builder_lib.h:1126-1166)
    int4 *node_rec;      // [2N]  per node {rp0, deg, j0, j1}{j2, j3, -, -}: CSR row start, in-degree and
                         //       its first four sources, so one 32-B read feeds the whole gather
    float *dinv;         // [N]   GCN normaliser     1/sqrt(1 + in_degree)
    float4 *gcoef;       // [N]   GCN: dinv_v * dinv_j of the four inline sources, 0 past the degree (launch_gcn_coef; valid when
                         //       the workspace says so: written lazily, in front of the first layer-wise GCN aggregate of a batch)
    float *amp;          // [N]   PNA amplification  log(max(d,1)+1)/delta
    float *att;          // [N]   PNA attenuation    delta/log(max(d,1)+1)
    int32_t *tile_first; // [T+1] first node of node-tile t; tiles are cut at graph boundaries
    int32_t *tile_edge;  // [T+1] first CSR slot of the tile (= edge_ptr of its first graph, clamped)
    int32_t *tile_graph; // [T+1] index of the graph that starts at tile_first[t] (B past the end)
    int32_t *node_graph; // [N] index of the graph every node row belongs to (written by graph prep when the model's last
                         //     conv layer can pool in its GEMM epilogue: launch_linear(..., PoolEpilogue); nullptr otherwise)
    int4 *agg_cut;       // [agg_cut_n + 1] row-balanced ranges of the gather-aggregate kernels' workgroups: {first row of range b
                         //       = floor(b N / n), start row / first CSR slot / index of the graph that row belongs to}; written
                         //       by graph prep when agg_cut_n > 0 (a power of two: the ring kernel's grid on this device)
    int32_t agg_cut_n;
    int32_t *stage_cut;  // [stage_cut_n + 1] tile ranges of the conv-stack kernels' workgroups as WHOLE stages of the batch's global
                         //       greedy stage list (graph prep, round 5): workgroup b walks tiles [stage_cut[b], stage_cut[b + 1]) -- every
                         //       workgroup the same number of stages (+- 1), no ragged last stage; valid when stage_cut_n > 0 (= the grid)
    int32_t stage_cut_n, stage_cut_cap;
    int32_t max_graph_nodes_hint; // caller's promise (0 = unknown); validated on device by prep
    int32_t promise_graphs;       // ... for graphs [0, promise_graphs) (the rest: the caller's "large segment")
    int32_t large_n, large_e;     // node / edge offset the caller named for graph promise_graphs (-1: no large segment);
                                  // graph prep flags the batch when they disagree with node_ptr / edge_ptr
    int32_t tile_lo;              // first node tile the gather-aggregate kernels walk (0; the large segment's first tile)
    int32_t *err;        // [1]   != 0 when the batch was malformed
    int32_t *err_host_dev; // device-visible address of the host-mapped copy of "flagged" (nullptr: none)
    const int32_t *node_ptr; // [B+1] caller's graph_node_ptr (device): read by graph prep ONLY
    int32_t *graph_ptr;      // [B+1] the same clamped to [0, N] by graph prep: what every later kernel reads, so
                             //       that a malformed node_ptr cannot send a pooling loop out of the buffers
    int32_t num_graphs, num_nodes, num_edges;
    int32_t tile_rows;   // target rows per tile
    int32_t num_tiles;
};

// bit of BatchTables::err set by a kernel that multiplies on 16-bit pieces (math modes 2 / 3) when it produced a non-finite
// value: the overflow contract of the reduced modes (gnnb_device.h: RangeProbe), reported as GNNB_ERR_RANGE
constexpr int GNNB_FLAG_RANGE = 64;
constexpr int GNNB_G2_STAGE_ROWS = 64;     // rows per stage of the fused 2-layer GCN kernel (4 MFMA units)
constexpr int GNNB_G2_STAGE_ROWS_BF6 = 48; // ... in the opt-in bf16x6 math mode (3 units)
// rows per stage of the transform-first 2-layer GCN kernel k_gcn2_zf for input width f0 and the promised graph size (176 or 96)
int zf_stage_rows(int f0, int promise);
long gcn2_zf_tile_capacity(int f0, int promise); // node tiles it can walk in one launch

// One tuning knob: a relaxed atomic int.  The knobs are process-wide and may be set (gnnb_set_option) while other threads launch
// -- round-5 review: "35+ knobs, unsynchronised".  Every read and write is now a single atomic access: no data race; a launcher that
// reads a knob twice may see two values, which only ever selects between kernels that give the same results (the one knob that
// changes results, "math", is captured per model: gnnb_model_desc::math, launch_math()).
struct OptInt {
    std::atomic<int> v;
    OptInt(int x) : v(x) {}
    operator int() const { return v.load(std::memory_order_relaxed); }
    OptInt &operator=(int x)
    {
        v.store(x, std::memory_order_relaxed);
        return *this;
    }
};
struct Options {
    OptInt tile_rows;    // node-tile granularity (rows; tiles are cut at graph boundaries)
    OptInt agg_lds_kb;   // LDS budget of the gather-aggregate kernel per workgroup (0 = all of the CU's / workgroups per CU)
    OptInt agg_ring_waves;   // waves per workgroup (0 = 16)
    OptInt agg_ring_slots;   // LDS stages in the ring
    OptInt agg_ring_wg_per_cu;
    OptInt agg_nt_store;     // non-temporal output stores
    OptInt agg_balance;      // 1 = the ring kernel's workgroups take row-balanced ranges (boundary graphs staged by both neighbours);
                          //     0 = whole-graph runs (default: measured faster, 15.2 vs 15.9 us at BASELINE config 2 -- the 3 % of extra
                          //     reads cost the per-CU memory pipe more than the +-8 % row imbalance)
    OptInt gemm_variant;  // 0 = register-resident weights when eligible (default), 1 = always the LDS-tiled kernel
    OptInt gemm_max_wg_per_cu;
    OptInt gemm_dma;      // 1 = large-K GEMM through LDS-DMA when every segment is plain (default)
    OptInt gemm_wlds;     // 1 = K, N in {64,128}, no skip operand: weights-in-LDS, barrier-free kernel (default); 0 = register-resident weights
    OptInt gemm_wlds_slots; // ... its ring depth per wave (capped by what fits beside W in LDS)
    OptInt fuse_narrow;   // 1 = aggregate + update of a narrow-input (F_in <= 32) GCN/GIN layer in one kernel
    OptInt first_ring;    // ... 1 = in ring form (k_conv_first: graphs staged in LDS, all columns from one stage; default), 0 = inside
                       //     k_linear_reg's A stage (round 3)
    OptInt fuse_zf;       // 1 = a 2-layer fp32 GCN stack takes k_gcn2_zf (last layer transformed before it is aggregated, 96-row
                       //     stages) instead of k_gcn2_fused (default); needs fuse_gcn2
    OptInt large_fork;    // a batch's large segment: 2 = through k_conv_rows on the caller's stream behind the stack kernel (default:
                       //     measured best with batches in flight: C3t 15.0 M graphs/s), 1 = k_conv_rows on a forked stream (shortest
                       //     single forward, 13.5 M in the pipeline: the stack kernels leave no register space for a co-resident
                       //     wave, so the fork only reorders), 0 = through the big layer-by-layer kernels (14.2 M)
    OptInt zf_shape;      // k_gcn2_zf: 0 = two 8-wave workgroups per CU, 96-row stages; 1 = one 16-wave workgroup, 176-row stages;
                       //     2 = shape 1 wherever it exists (input widths up to 16), else shape 0 (default)
    OptInt fuse_gcn2;     // 1 = fused 2-layer GCN stack when the model and the max_graph_nodes hint allow it (k_gcn2_fused), 0 = layer by layer
    OptInt fuse_head;     // 1 = pooling + MLP head in one kernel when it fits (default)
    OptInt fuse_pool;     // 1 = global pooling in the epilogue of the last conv layer's GEMM where that GEMM has one (GraphSAGE's
                       //     large-K segmented GEMM; default), 0 = separate pooling pass
    OptInt head_small;    // 1 = readout on a pooled matrix with the small-footprint kernel that co-resides with the
                       //     conv-stack kernel of the next batch in flight (default); 0 = weights-in-LDS kernel
    OptInt head_split;    // 1 = layer-wise models: pooling pass + small readout instead of the one-launch pooling+MLP kernel
    OptInt math;          // 0 = fp32 MFMA everywhere (default); 1 = the wide update of the fused GCN stack and the
                       //     K <= 128 GEMMs as six bf16 MFMA products of an exact 3-way split of both operands
                       //     (fp32-equivalent, opt-in)
    OptInt gemm_tail_split; // k_linear_dma's last, partial round of tiles: 2 = cut along K into equal runs over all resident
                         // workgroups, parts added up by the last one at each tile (stream-K; default), 1 = handed out as
                         // row slices (bit-identical to 0), 0 = whole tiles
    OptInt pna_fold_lin;    // 1 = PNA's `lin` folded into its post-NN at upload: one 13F-wide GEMM per layer (default); 0 = two GEMMs
    OptInt pna_classes;     // 1 = PNA under a max_degree promise <= 15: rows sorted by degree, 5F-wide GEMM with per-class weights (default)
    OptInt fold_skip;       // 1 = GraphSAGE: a middle layer's skip connection folded into the root weights (Wr + I) instead of read as an operand (default)
    OptInt sage_first_mean; // 1 = GraphSAGE: the narrow first layer also forms the NEXT layer's mean aggregate from its output rows while they
                         //     are in LDS (k_sage_first_mean; needs the max_graph_nodes promise; default); 0 = k_conv_first + aggregate kernel
    OptInt pna_first;       // 1 = a PNA layer with a narrow input (F <= 12: the first) as ONE kernel -- pre-NN, aggregate, scalers, post-NN --
                         //     when the max_graph_nodes promise lets whole graphs be staged (k_pna_first; default); 0 = four launches
    OptInt pna_pagg;        // 1 = a full-width PNA layer under the degree promise + the max_graph_nodes promise: pre-NN product and aggregate in
                         //     one kernel, p never in HBM (k_pna_pagg; default); 0 = GEMM + k_aggregate_ring<PNA>
    OptInt stage_cut;       // 1 = k_gcn2_fused's workgroups take whole stages of the batch's global greedy stage list (graph prep plans the
                         //     cuts: k_stage_cut -- the kernel alone 232 -> 215 us at BASELINE config 3, but the planner's latency costs the
                         //     three-stream pipeline more than that: opt-in); 0 = equal tile counts (default)
    OptInt zf_head;         // 1 = k_gcn2_zf runs the MLP head on the graphs it pooled (conv stack + pooling + head in one launch: measured
                         //     slower than the separate readout); 0 = separate readout launch (default)
    OptInt agg_form;        // gather-aggregate kernel: 0 = LDS ring (k_aggregate_ring), 1 = register gather (k_aggregate_rg: no LDS, no
                         // barrier; widths 64 / 128 / 256, kinds GCN / SUM / MEAN / SIMPLE / PNA; anything else falls back to the ring),
                         // 2 = register gather for PNA only.  Default 0: DESIGN 3.2 (a wash at config 2, slower inside config 4's step)
    OptInt agg_rg_r;        // ... row-instructions in flight per wave and batch (0 = 1)
    OptInt agg_rg_wgs;      // ... 256-thread workgroups per CU, resident or not (0 = 32 at one row-instruction per batch)
    OptInt agg_rg_flags;    // ... bit 0: sources past the degree are not loaded (exec-masked; default) instead of aliasing the row itself;
                         //     bit 1: the run pre-touched line by line (slower); bits 2, 3: ablations (GCN, w = 128, one row-instruction)
    OptInt prep_group;      // graphs per wave of the molecule-path graph prep (k_graph_prep<64, G>: fetches batched): 4 (default; batches of >= 2047 graphs), 1
    OptInt head_pairs;      // readout on a pooled matrix (k_head_small): 1 = operands in pairs, 72 registers: shares a SIMD with a graph-prep wave
                         //     beside the stack kernel (default); 0 = four slices in flight, 82 registers: faster with one batch in flight
    OptInt guest_prep;      // 1 = gnnb_forward_prepared_prep_next runs the next batch's graph prep inside k_gcn2_zf where the batch is
                         //     eligible (default); 0 = always as k_graph_prep behind the forward
};
Options &options();

// The math mode of the launches the CALLING THREAD issues (0 fp32 MFMA, 1 bf16x6, 2 bf16x3, 3 f16x3: include/gnnb_hip.h).
// Inside a model's forward or its workspace's graph prep: the model's own mode (gnnb_model_desc::math, captured by
// gnnb_model_create) -- two designs of different precision in one process, or another thread calling gnnb_set_option,
// never change each other's arithmetic (reference: precision is baked into the generated design, model.h.jinja:38-62).
// Outside (the stand-alone gnnb_linear / gnnb_aggregate entries) and for models created with math = -1: the process-wide
// option as it stands at the launch.  Every launcher reads launch_math(), never options().math.
int launch_math();
// ... and the flag word of the workspace whose forward the calling thread is in ({nullptr, nullptr} outside: the stand-alone
// entries have no workspace and report nothing): where the reduced modes' kernels that take no BatchTables (k_linear_dma) drop
// GNNB_FLAG_RANGE
struct FlagWord {
    int32_t *err, *err_host;
};
FlagWord launch_flag_word();
struct MathScope { // entry points: `MathScope scope(ws->desc.math, ws->t.err, ws->t.err_host_dev);`
    int prev;
    FlagWord prev_flag;
    explicit MathScope(int model_math, int32_t *err = nullptr, int32_t *err_host = nullptr);
    ~MathScope();
    MathScope(const MathScope &) = delete;
    MathScope &operator=(const MathScope &) = delete;
};

// Arguments of one graph prep (k_graph_prep; round 6: also of the prep the readout kernel runs for its stream's next batch, k_readout.hip)
struct PrepParams {
    const int2 *coo;
    const int32_t *node_ptr, *edge_ptr;
    int B, N, E;
    int32_t *row_ptr, *col, *eid;
    int4 *node_rec;
    float *dinv, *amp, *att;
    float delta;
    int32_t *tile_first, *tile_edge, *tile_graph, *graph_ptr;
    int tile_rows, num_tiles, max_graph_nodes_hint, promise_graphs, large_n, large_e, drop_self;
    int32_t *err, *err_host;
    int4 *agg_cut;
    int cut_log2;
    int32_t *node_graph;
};
// A graph prep on offer to the launches of a forward on this thread (gnnb_forward_prepared_prep_next): the launcher that runs it
// inside its kernel sets `taken` (k_head_small, k_readout.hip), otherwise the caller launches k_graph_prep
struct GuestPrep {
    const PrepParams *params;
    bool taken;
};
GuestPrep *&guest_prep_slot(); // thread-local; nullptr = nothing on offer
PrepParams make_prep_params(const int32_t *coo, const int32_t *node_ptr, const int32_t *edge_ptr, const BatchTables &t, float pna_delta,
                            int drop_self_loops);
hipError_t launch_graph_prep(const PrepParams &p, hipStream_t s);
// drop_self_loops: edges (v, v) are not entered into the tables (GCN: PyG's add_remaining_self_loops)
hipError_t launch_graph_prep(const int32_t *coo, const int32_t *node_ptr, const int32_t *edge_ptr,
                             BatchTables &t, float pna_delta, int drop_self_loops, hipStream_t s);

hipError_t launch_aggregate(const BatchTables &t, int kind, const float *x, const float *selfq,
                            float *out, int width, float eps, hipStream_t s);
// GraphSAGE's narrow first layer (k_conv_first's [mean | x] form) + the next layer's mean aggregate of its output rows, one kernel
// (k_first_mean.hip): y [N, Nout] and mean_out [N, Nout].  hipErrorNotSupported (nothing launched) -> launch_conv_first + aggregate
hipError_t launch_sage_first_mean(const BatchTables &t, const float *x, int F, const float *w, int ldw, const float *bias, float *y,
                                  float *mean_out, int Nout, int act, hipStream_t s);
// A narrow-input PNA layer in one kernel (k_pna_first.hip): y = act([x | A | amp A | att A] . w^T + bias) with A = the four
// statistics of W_pre [x_i || x_j] + b_pre over every node's sources; wpre [F][2F], w [Nout][ldw >= 13 F] (post-NN with `lin`
// folded in); t.amp / t.att must have been prepared with the model's delta.  hipErrorNotSupported (nothing launched) -> the
// layer-by-layer kernels
hipError_t launch_pna_first(const BatchTables &t, const float *x, int F, const float *wpre, const float *bpre, const float *w, int ldw,
                            const float *bias, float *y, int Nout, int act, hipStream_t s);
// PNA: out [N, 4F] = max | min | mean | std over every node's sources of p_j = Wb x_j, p kept on chip (k_pna.hip); no destination
// term (the degree-class form).  hipErrorNotSupported (nothing launched) -> p GEMM + launch_aggregate(GNNB_AGG_PNA)
hipError_t launch_pna_pagg(const BatchTables &t, const float *x, int F, const float *wb, int ldw, float *out, hipStream_t s);
// the register-gather form (k_aggregate_rg.hip); hipErrorNotSupported (nothing launched) -> the ring form
hipError_t launch_aggregate_rg(const BatchTables &t, int kind, const float *x, const float *selfq, float *out, int width,
                               float eps, hipStream_t s);
// t.gcoef from t.node_rec / t.dinv: what launch_aggregate(GNNB_AGG_GCN) reads (call once per prepared batch)
hipError_t launch_gcn_coef(const BatchTables &t, hipStream_t s);
// workgroups the ring-form gather-aggregate kernel launches on the current device (CUs x agg_ring_wg_per_cu): what graph
// prep cuts its row-balanced ranges for
int aggregate_ring_grid();
// GINE: out_i = (1 + eps) x_i + sum_j relu(x_j + eterm[edge]); eterm [E, width] in COO order
hipError_t launch_aggregate_edges(const BatchTables &t, const float *x, const float *eterm, float *out, int width,
                                  float eps, hipStream_t s);

// Fused readout: global pooling + the whole MLP head in one launch (16 graphs per workgroup).
struct HeadArgs {
    const float *w[8];
    const float *b[8];
    int32_t dims[9]; // dims[0] = num_pools * d, dims[i+1] = output width of linear i
    int32_t nlin;
};
struct GemmArgs {
    const float *a[4];
    const float *rs[4];
    int32_t lda[4];
    int32_t k[4];
    int32_t koff[4];   // column offset of segment s inside W
    int32_t cpre[5];   // prefix sums of ceil(k/32)
    int32_t avec[4];   // segment may use 16-byte loads for A
    int32_t wvec[4];   // ... for W
    int32_t nseg;
};
// aggregate + dense update of a narrow-input layer in one launch; hipErrorNotSupported -> caller
// runs launch_aggregate + launch_linear
hipError_t launch_conv_gather(const BatchTables &t, int agg_kind, float eps, const float *x, int lda,
                              int K, const float *w, int ldw, const float *bias, const float *skip,
                              float *y, int N, int act, hipStream_t s, int cat = 0);

// Global pooling folded into the LAST conv layer's GEMM epilogue (reference compute_global_graph_pooling,
// templates/model.cpp.jinja:413-449, global_*_pool gnn_builder_lib.h:2709-2803): the [N, d] output of the layer is never
// written.  Rows are pooled in 32-row blocks: a graph that lies inside one block is finished there (-> pooled), the pieces
// of a graph that crosses block boundaries go to `part` and launch_pool_combine adds them up in row order (deterministic).
struct PoolEpilogue {
    const int32_t *node_graph = nullptr; // BatchTables::node_graph
    const int32_t *graph_ptr = nullptr;  // [B + 1]
    float *pooled = nullptr;             // [B, np, N]
    float2 *part = nullptr;              // [ceil(M / 32), 2, N] {sum, max} of a block's first / last open piece
    int32_t num_graphs = 0, np = 0, pools[3] = {0, 0, 0};
};
// Stream-K tail of k_linear_dma (k_gemm.hip): q chunks per run; part[2 * workgroup + segment][4 waves][64 x 64 lanes]
// accumulators, cnt[first workgroup of a shared tile] arrival counters (zero between launches).  q = 0: off.
struct StreamK {
    int q = 0;
    float *part = nullptr;
    int *cnt = nullptr;
};
// a workspace's own stream-K scratch: stream_k_scratch_bytes() of device memory whose counter part (the tail) the owner has
// zeroed; stream_k_scratch_at(base) names its two pieces.  Handed to launch_linear(..., sk_owned); without one (the
// standalone gnnb_linear entry) launch_linear keeps a scratch per (device, stream) and never uses it under graph capture.
size_t stream_k_scratch_bytes();
StreamK stream_k_scratch_at(void *base);
hipError_t stream_k_scratch_init(void *base, hipStream_t s); // counters zeroed, guard pattern written (in stream order)
hipError_t stream_k_scratch_init_sync(void *base);           // the same with synchronous memsets (workspace creation)
// diagnostics: 1 = the scratch's counters are all zero and the guard region behind them is whole, 0 = not, -1 = read-back failed.
// owned == nullptr: the standalone scratch of (current device, s), 1 when there is none yet.  Synchronises s.
int stream_k_guard_intact(const StreamK *owned, hipStream_t s);
constexpr int GNNB_DEG_CLASSES = 16; // in-degrees 0 .. 15 get a class each (0: PNA's scalers of degree 1, but no messages); a larger promise keeps the general form
constexpr int GNNB_DEG_MAX = GNNB_DEG_CLASSES - 1;
// the batch's rows sorted into degree classes (k_misc.hip); work = 256 x 16 ints, perm = max_tiles * 128, tile_cls = max_tiles
hipError_t launch_degree_classes(const BatchTables &t, int promise, int32_t *work, int32_t *perm, int32_t *tile_cls, int max_tiles,
                                 hipStream_t s);
// Row classes of k_linear_dma (PNA under a degree promise, k_gemm.hip): perm[position in the class-sorted space] = row or -1,
// tile_cls[128-row tile of that space] = class, whose weight matrix starts w_stride floats after the previous one's.
struct RowClasses {
    const int32_t *perm = nullptr;
    const int32_t *tile_cls = nullptr;
    long w_stride = 0;
    int bias_stride = 0; // floats between the classes' biases (0: one bias)
};
// pe != nullptr: y is not written; returns hipErrorNotSupported (nothing launched) when the GEMM shape has no pooling
// epilogue -- the caller then runs the plain GEMM + a pooling pass
hipError_t launch_linear(const GemmArgs &g, const float *w, int ldw, const float *bias,
                         const float *skip, float *y, int M, int N, int act, hipStream_t s, const PoolEpilogue *pe = nullptr,
                         const RowClasses *rc = nullptr, const StreamK *sk_owned = nullptr);
// the same narrow-input layer in ring form (k_first.hip): whole graphs staged in LDS once for ALL N <= 256 output columns;
// F = width of x, K = F or (cat = F) 2 F.  hipErrorNotSupported -> launch_conv_gather's k_linear_reg form
hipError_t launch_conv_first(const BatchTables &t, int agg_kind, float eps, const float *x, int F, int K, const float *w,
                             int ldw, const float *bias, float *y, int Nout, int act, hipStream_t s, int cat);
// the pieces of graphs that cross 32-row blocks -> pooled (and zeros for empty graphs); after launch_linear(..., pe)
hipError_t launch_pool_combine(const PoolEpilogue &pe, int M, int N, hipStream_t s);

// returns hipErrorNotSupported when the head does not fit the fused kernel (caller falls back)
hipError_t launch_pool_mlp(const float *x, const int32_t *node_ptr, int num_graphs, int d,
                           const int32_t *pools, int num_pools, const HeadArgs &head, int act,
                           float *out, hipStream_t s, const float *prepooled = nullptr);

// Whole 2-layer GCN conv stack + pooling in one persistent launch (graphs staged once in LDS):
// x -> agg -> W0 -> act -> agg -> W1 -> act -> pooled [B, np*h1].  hipErrorNotSupported when the
// model / batch does not qualify (caller runs the layer-by-layer path).
// layers between the first and the last of a fused GCN stack (width h0 -> h0): layer l's weight at wmid + (l - 1) *
// mid_stride floats, its bias at bmid + (l - 1) * bmid_stride; nl = number of conv layers (2: no middle layers)
// gin: a GIN stack instead -- wmid / bmid then address ALL hidden x hidden matrices in execution order (index 0 = layer 0's
// second linear, 2l - 1 / 2l = layer l's first / second linear), eps = GINConv's (1 + eps) self weight
struct G2Deep {
    int nl = 2;
    const float *wmid = nullptr, *bmid = nullptr;
    long mid_stride = 0, bmid_stride = 0;
    int skip = 0;
    int gin = 0;
    float eps = 0.0f;
};
long gcn2_fused_tile_capacity();
int gcn2_fused_grid(int num_tiles);       // workgroups k_gcn2_fused launches for that many node tiles (CUs x 2, at most one per tile)
int gcn2_fused_tile_window();             // tiles a workgroup's run may span (its tile-table window in LDS)
// the conv-stack kernels' runs as whole stages of the global greedy stage list (k_plan.hip): cut [G + 2] (cut[G + 1] = valid)
int stage_cut_levels(int max_tiles);
hipError_t launch_stage_cut(const int32_t *tile_first, int num_tiles, int num_nodes, int cap, int G, int tcap, int32_t *scratch,
                            int32_t *cut, hipStream_t s);
hipError_t launch_gcn2_fused(const BatchTables &t, const float *x, int f0, const float *w0, const float *b0,
                             int h0, const float *w1, const float *b1, int h1, int act,
                             const int32_t *pools, int num_pools, float *pooled, hipStream_t s, const G2Deep &deep = G2Deep{});

// The same stack for exactly two GCN layers in fp32 math, last layer transformed before it is aggregated (k_stack_zf.hip)
// head != nullptr: the MLP head (activation = `act`) runs inside the kernel too when its shape allows (*head_fused says so):
// out [B, mlp_out] is then complete and the caller launches no readout for these graphs
hipError_t launch_gcn2_zf(const BatchTables &t, const float *x, int f0, const float *w0, const float *b0,
                          int h0, const float *w1, const float *b1, int h1, int act,
                          const int32_t *pools, int num_pools, float *pooled, hipStream_t s, const float *w1_frag_order = nullptr,
                          const HeadArgs *head = nullptr, const HeadArgs *head_dev = nullptr, float *head_out = nullptr,
                          bool *head_fused = nullptr);
hipError_t launch_gcn2_zf_head(const BatchTables &t, const float *x, int f0, const float *w0, const float *b0,
                          int h0, const float *w1, const float *b1, int h1, int act,
                          const int32_t *pools, int num_pools, float *pooled, hipStream_t s, const float *w1_frag_order = nullptr,
                          const HeadArgs *head = nullptr, const HeadArgs *head_dev = nullptr, float *head_out = nullptr,
                          bool *head_fused = nullptr); // k_stack_zf_head.hip: the kernels WITH the MLP-head tail  (head: host copy for the shape checks; head_dev: the same in device memory, what the kernel reads)

// One GCN / GIN conv layer for node rows [row_lo, N) in the small-footprint form that co-resides with the stack kernels
// (k_conv_rows.hip: the large segment of a batch).  hipErrorNotSupported: widths beyond 128 or another conv type.
hipError_t launch_conv_rows(const BatchTables &t, int conv_type, const float *x, int K, const float *w1, const float *b1,
                            const float *w2, const float *b2, int Nout, const float *skip, float *y, int row_lo, int act,
                            float eps, hipStream_t s);

// dst = src put on the ap_fixed<W, I> grid (truncate, wrap); dst may alias src
hipError_t launch_quantize(const float *src, float *dst, size_t n, int W, int I, hipStream_t s);

hipError_t launch_output_activation(float *out, int num_graphs, int n, int kind, hipStream_t s);

hipError_t launch_global_pool(const float *x, const int32_t *node_ptr, int num_graphs, int d,
                              const int32_t *pools, int num_pools, float *out, hipStream_t s);

} // namespace gnnb
