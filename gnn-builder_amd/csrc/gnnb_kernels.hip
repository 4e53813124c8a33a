// gnnb_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels of the GNNBuilder hot path.
//
//   k_graph_prep   COO -> CSR-by-destination + degree scalers + node tiles   (HBM / latency bound)
//   k_aggregate    gather -> segmented reduce per destination row            (HBM bound: THE roofline kernel)
//   k_linear*      multi-segment X.W^T + bias + skip + activation on fp32 MFMA (matrix-core bound):
//                  k_linear_wlds / k_linear_reg (K <= 128), k_linear_dma (large K), k_linear (irregular shapes)
//   k_global_pool, k_pool_mlp, k_head_small   per-graph add / mean / max readout + MLP head   (HBM bound)
//   k_gcn2_fused   whole GCN / GIN conv stack + pooling in one persistent kernel, graphs staged in LDS (matrix-core bound)
//
// Wavefront = 64 lanes everywhere.  Reference semantics are cited per kernel
// (paths relative to the reference repository root).
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "gnnb_internal.h"

namespace gnnb {

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised first.
// The attribute is per device and per kernel: remembered here per (device, kernel) under a lock, so that launches
// from several host threads or on several devices of one process each get it, and the runtime call (a few
// microseconds of host time) is not paid on every launch.
static hipError_t ensure_dynamic_lds(const void *kern, size_t lds)
{
    if (lds <= 64 * 1024)
        return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[std::make_pair(dev, kern)];
    if (have >= lds)
        return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
        have = lds;
    return e;
}

static constexpr int WG = 256; // 4 wavefronts

// compile-time integer tag (generic lambdas dispatch on it)
template <int V>
struct IntTag {
    static constexpr int value = V;
};

// Diagnostic build only (-DGNNB_PROBE, tools/probe_agg.py): per-workgroup phase stamps.  The
// product library is built without it and executes no stamp.
#ifdef GNNB_PROBE
__device__ unsigned long long g_probe[16 * 8192];
#define GNNB_STAMP(slot)                                                                   \
    do {                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 8192) {                                       \
            g_probe[blockIdx.x * 8 + 2 * (slot)] = wall_clock64();                         \
            g_probe[blockIdx.x * 8 + 2 * (slot) + 1] = clock64();                          \
        }                                                                                  \
    } while (0)
#define GNNB_STAMP_END(slot)                                                               \
    do {                                                                                   \
        __builtin_amdgcn_s_waitcnt(0); /* drain this wave's stores first */                \
        GNNB_STAMP(slot);                                                                  \
    } while (0)
#else
#define GNNB_STAMP(slot) do { } while (0)
#define GNNB_STAMP_END(slot) do { } while (0)
#endif

// =====================================================================================
// graph prep
// =====================================================================================
// Reference: compute_degree_tables + compute_neighbor_tables
// (gnnbuilder/gnn_builder_lib/gnn_builder_lib.h:1051-1083, :1086-1124): in-degree, exclusive
// prefix sum, stable counting sort of sources by destination.  The reference runs this
// serially per graph; here ONE WAVEFRONT owns one graph of the batch: lane = destination
// node, the graph's (few dozen) edges are scanned by register broadcasts, so the sort is
// stable by construction and needs no atomics.  Edges of a graph are contiguous
// (edge_ptr), so the batch-global CSR segment of graph g starts at edge_ptr[g].
// The graph's edges live in REGISTERS: lane l of the wave holds edge 64c+l of chunk c, and the
// scan over edges broadcasts one edge at a time with v_readlane (scalar index) -- no LDS, no
// per-edge memory latency.  Graphs of up to 64*PREP_REG_CHUNKS edges take this path; larger ones
// re-read their edge list from global memory (L2) chunk by chunk.
static constexpr int PREP_REG_CHUNKS = 4;

// General path (any graph size): lane = destination node, edges scanned one at a time.

// Batch-validation flag: the authoritative word lives in device memory (read and reset by gnnb_workspace_check); a
// copy of "something was flagged" is also dropped into a host-mapped word, which the NEXT entry call on the workspace
// reads without synchronising (lazy detection for callers that never call the check).
__device__ __forceinline__ void flag_batch(int32_t *err, int32_t *err_host, int bits)
{
    atomicOr(err, bits);
    if (err_host)
        *reinterpret_cast<volatile int32_t *>(err_host) = bits;
}

__device__ void prep_graph_scan(
    const int2 *__restrict__ coo, int n0, int n1, int e0, int e1, int32_t *__restrict__ row_ptr,
    int32_t *__restrict__ col, int32_t *__restrict__ eid, int4 *__restrict__ node_rec, float *__restrict__ dinv,
    float *__restrict__ amp, float *__restrict__ att, float delta, int drop_self, int32_t *__restrict__ err,
    int32_t *__restrict__ err_host)
{
    const int lane = threadIdx.x & 63;

    const int ne = e1 - e0;
    const int nchunks = (ne + 63) >> 6;
    const bool inreg = nchunks <= PREP_REG_CHUNKS; // wave-uniform
    bool bad = false;

    // an edge that leaves its graph is an error: it is neutralised (dst = -1 never matches, src
    // clamped) so that later gathers stay in range
    auto fetch = [&](int c, int &es, int &ed) {
        const int i = c * 64 + lane;
        es = n0;
        ed = -1;
        if (i < ne) {
            const int2 e = coo[e0 + i];
            if (e.x < n0 || e.x >= n1 || e.y < n0 || e.y >= n1)
                bad = true;
            else if (!(drop_self && e.x == e.y)) { // GCN: an explicit self loop is not an edge (PyG add_remaining_self_loops)
                es = e.x;
                ed = e.y;
            }
        }
    };
    int rs[PREP_REG_CHUNKS], rd[PREP_REG_CHUNKS];
#pragma unroll
    for (int c = 0; c < PREP_REG_CHUNKS; c++) {
        rs[c] = n0;
        rd[c] = -1;
        if (inreg && c < nchunks)
            fetch(c, rs[c], rd[c]);
    }

    int base = e0;
    for (int c0 = n0; c0 < n1; c0 += 64) {
        const int v = c0 + lane;
        const bool active = v < n1;
        // ---- in-degree of node v: scan the edges, one broadcast per edge
        int cnt = 0;
        if (inreg) {
#pragma unroll
            for (int c = 0; c < PREP_REG_CHUNKS; c++) {
                if (c < nchunks) {
                    const int m = min(64, ne - c * 64);
                    for (int i = 0; i < m; i++)
                        cnt += (__builtin_amdgcn_readlane(rd[c], i) == v) ? 1 : 0;
                }
            }
        } else {
            for (int c = 0; c < nchunks; c++) {
                int es, ed;
                fetch(c, es, ed);
                const int m = min(64, ne - c * 64);
                for (int i = 0; i < m; i++)
                    cnt += (__builtin_amdgcn_readlane(ed, i) == v) ? 1 : 0;
            }
        }
        if (!active)
            cnt = 0;
        // wave-wide inclusive scan of the in-degrees
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += t;
        }
        const int start = base + incl - cnt;
        if (active) {
            row_ptr[v] = start;
            dinv[v] = 1.0f / sqrtf(1.0f + (float)cnt);
            const int dcl = cnt < 1 ? 1 : cnt; // gnn_builder_lib.h:1972-1982
            const float logd = logf((float)(dcl + 1));
            if (delta > 0.0f) { // (delta <= 0: the model has no PNA layer, the scalers are not needed)
                amp[v] = logd / delta;
                att[v] = delta / logd;
            }
        }
        // ---- stable fill: edges are visited in COO order; the first four sources also go into
        // the node record
        int pos = start;
        int jf[4] = {v, v, v, v};
        auto put = [&](int src, int edge) {
            const int q = pos - start;
            if (q == 0) jf[0] = src;
            else if (q == 1) jf[1] = src;
            else if (q == 2) jf[2] = src;
            else if (q == 3) jf[3] = src;
            eid[pos] = edge; // COO row of the CSR slot (compute_neighbor_and_edge_index_tables, gnn_builder_lib.h:1126-1166)
            col[pos++] = src;
        };
        if (inreg) {
#pragma unroll
            for (int c = 0; c < PREP_REG_CHUNKS; c++) {
                if (c < nchunks) {
                    const int m = min(64, ne - c * 64);
                    for (int i = 0; i < m; i++) {
                        const int d = __builtin_amdgcn_readlane(rd[c], i);
                        const int sc = __builtin_amdgcn_readlane(rs[c], i);
                        if (d == v)
                            put(sc, e0 + c * 64 + i);
                    }
                }
            }
        } else {
            for (int c = 0; c < nchunks; c++) {
                int es, ed;
                fetch(c, es, ed);
                const int m = min(64, ne - c * 64);
                for (int i = 0; i < m; i++) {
                    const int d = __builtin_amdgcn_readlane(ed, i);
                    const int sc = __builtin_amdgcn_readlane(es, i);
                    if (d == v)
                        put(sc, e0 + c * 64 + i);
                }
            }
        }
        if (active) {
            node_rec[2 * (size_t)v] = make_int4(start, cnt, jf[0], jf[1]);
            node_rec[2 * (size_t)v + 1] = make_int4(jf[2], jf[3], 0, 0);
        }
        base += __shfl(incl, 63, 64);
    }
    if (bad)
        flag_batch(err, err_host, 4);
}


// Fast path (graphs of <= 256 nodes and <= 256 edges, i.e. every molecule): lanes hold EDGES.
// One loop over the graph's destination nodes: ballot(dst == v) gives, in a single instruction,
// the in-degree of v (popcount) and the rank of every edge among v's in-edges (popcount of the
// lower lanes) -- stable, because lanes are in COO order.  Starts come from a wave prefix sum over
// node lanes, and `col` is then written by ONE scatter per 64 edges instead of a divergent
// store per edge.  n iterations of ~8 instructions replace 2e iterations of a dependent chain.
static constexpr int PREP_FAST_EDGES = 256; // 4 edge chunks

// PREP_FAST_NODES: 256 (4 node chunks of 64 lanes) in general, 64 when the caller promises graphs of <= 64
// nodes -- 4 KB of LDS per workgroup instead of 16 KB, so that graph prep of the next batch fits on a CU
// beside two workgroups of the conv-stack kernel and the readout of the previous one.
template <int PREP_FAST_NODES>
__global__ __launch_bounds__(WG) void k_graph_prep(
    const int2 *__restrict__ coo, const int32_t *__restrict__ node_ptr,
    const int32_t *__restrict__ edge_ptr, int B, int N, int E, int32_t *__restrict__ row_ptr,
    int32_t *__restrict__ col, int32_t *__restrict__ eid, int4 *__restrict__ node_rec, float *__restrict__ dinv,
    float *__restrict__ amp, float *__restrict__ att, float delta,
    int32_t *__restrict__ tile_first, int32_t *__restrict__ tile_edge, int32_t *__restrict__ tile_graph,
    int32_t *__restrict__ graph_ptr, int tile_rows, int num_tiles, int max_graph_nodes_hint, int drop_self,
    int32_t *__restrict__ err, int32_t *__restrict__ err_host)
{
    __shared__ int32_t s_first[WG / 64][PREP_FAST_NODES * 4]; // first four sources of every node
    // Highest wave priority: with batches in flight on several streams this kernel runs BESIDE the conv-stack kernel of
    // another batch (one wave slot per SIMD is left over there) and sits on its own stream's critical path -- 37-53 us
    // instead of 7 when it queues behind sixteen MFMA-issuing waves per CU.  It is a few hundred instructions per wave;
    // letting them issue first costs the big kernel nothing measurable (C2 step 55.5 -> 54.0 us).
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = blockIdx.x * (WG / 64) + wave;
    if (g > B)
        return;

    // ---- node tiles: tile_first[t] = min{ node_ptr[g'] : node_ptr[g'] >= t*tile_rows }
    {
        // clamped so that a malformed node_ptr (flagged below) cannot write out of range
        const int p = (g == B) ? N : min(max(node_ptr[g], 0), N);
        const int t_lo = (g == 0) ? 0 : max(min(max(node_ptr[g - 1], 0), N) / tile_rows + 1, 0);
        const int t_hi = (g == B) ? num_tiles : min(p / tile_rows, num_tiles);
        // edges are grouped by graph, so the CSR segment of graph g starts at edge_ptr[g]
        const int pe = (g == B) ? E : min(max(edge_ptr[g], 0), E);
        for (int t = t_lo + lane; t <= t_hi; t += 64) {
            tile_first[t] = p;
            tile_edge[t] = pe;
            tile_graph[t] = g;
        }
        if (lane == 0)
            graph_ptr[g] = p; // the clamped copy later kernels read
    }
    // Containment of malformed batches: whatever node_ptr / edge_ptr hold, every row in [0, N) leaves this
    // kernel with a record that later kernels can follow without leaving the buffers -- start and start + deg
    // inside [0, E], sources inside [0, N).  A graph's ranges are CLAMPED instead of rejected (any row r < N lies
    // in some pair node_ptr[g] <= r < node_ptr[g+1] when node_ptr runs from 0 to N; rows before node_ptr[0] or
    // after node_ptr[B] are given empty records by the last wave), only edges inside the clamped node range are
    // accepted, and the results of a flagged batch are unspecified but in range.
    auto empty_rows = [&](int r0, int r1) {
        for (int v = r0 + lane; v < r1; v += 64) {
            row_ptr[v] = 0;
            node_rec[2 * (size_t)v] = make_int4(0, 0, v, v);
            node_rec[2 * (size_t)v + 1] = make_int4(v, v, 0, 0);
            dinv[v] = 1.0f;
            if (delta > 0.0f) {
                amp[v] = logf(2.0f) / delta;
                att[v] = delta / logf(2.0f);
            }
        }
    };
    if (g == B) {
        const int first = node_ptr[0], last = node_ptr[B];
        if (lane == 0) {
            row_ptr[N] = E;
            if (last != N || edge_ptr[B] != E || first != 0 || edge_ptr[0] != 0)
                flag_batch(err, err_host, 1);
        }
        if (first > 0)
            empty_rows(0, min(first, N));
        if (last < N)
            empty_rows(max(last, 0), N);
        return;
    }

    GNNB_STAMP(0);
    int n0 = node_ptr[g], n1 = node_ptr[g + 1];
    int e0 = edge_ptr[g], e1 = edge_ptr[g + 1];
    if (n0 > n1 || e0 > e1 || n1 > N || e1 > E || n0 < 0 || e0 < 0) {
        if (lane == 0)
            flag_batch(err, err_host, 2);
        n0 = min(max(n0, 0), N);
        n1 = min(max(n1, 0), N);
        e0 = min(max(e0, 0), E);
        e1 = min(max(e1, 0), E);
        if (n0 >= n1)
            return; // covers no row
        if (e0 > e1)
            e1 = e0; // no usable edge range: the rows get empty records
    }
    const int n = n1 - n0, ne = e1 - e0;
    if (max_graph_nodes_hint > 0 && n > max_graph_nodes_hint && lane == 0)
        flag_batch(err, err_host, 8); // the caller's max_graph_nodes promise does not hold for this batch
    if (n > PREP_FAST_NODES || ne > PREP_FAST_EDGES) { // wave-uniform
        prep_graph_scan(coo, n0, n1, e0, e1, row_ptr, col, eid, node_rec, dinv, amp, att, delta, drop_self, err, err_host);
        return;
    }

    // ---- edges -> registers (lane l holds edge 64c + l); an edge that leaves its graph is an error
    // and is dropped (dst = -1 never matches)
    constexpr int EC = PREP_FAST_EDGES / 64, NC = PREP_FAST_NODES / 64;
    int es[EC], ed[EC], erank[EC];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < EC; c++) {
        es[c] = n0;
        ed[c] = -1;
        erank[c] = 0;
        const int i = c * 64 + lane;
        if (i < ne) {
            const int2 e = coo[e0 + i];
            if (e.x < n0 || e.x >= n1 || e.y < n0 || e.y >= n1)
                bad = true;
            else if (!(drop_self && e.x == e.y)) { // GCN: an explicit self loop is not an edge (see gnnb_hip.h)
                es[c] = e.x;
                ed[c] = e.y - n0; // local destination
            }
        }
    }
    GNNB_STAMP(1);
    // ---- one pass over destination nodes: degree of node v -> node lane (v & 63) of chunk v >> 6;
    // rank of each matching edge among v's in-edges -> that edge's lane
    int deg[NC];
#pragma unroll
    for (int q = 0; q < NC; q++)
        deg[q] = 0;
    const int nec = (ne + 63) >> 6;
    if (nec <= 1 && n <= 64) {
        // the common molecule case (<= 64 nodes, <= 64 edges): one chunk each, ~6 instructions per node
        for (int v = 0; v < n; v++) {
            const unsigned long long m = __ballot(ed[0] == v);
            if (ed[0] == v)
                erank[0] = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (lane == v)
                deg[0] = __popcll(m);
        }
    } else {
        for (int v = 0; v < n; v++) {
            int before = 0; // in-edges of v in earlier edge chunks
#pragma unroll
            for (int c = 0; c < EC; c++) {
                if (c < nec) {
                    const unsigned long long m = __ballot(ed[c] == v);
                    if (ed[c] == v)
                        erank[c] = before + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    before += __popcll(m);
                }
            }
#pragma unroll
            for (int q = 0; q < NC; q++)
                if ((v >> 6) == q && lane == (v & 63))
                    deg[q] = before;
        }
    }
    GNNB_STAMP(2);
    // ---- row starts: wave prefix sum over node lanes, chunk by chunk
    int start[NC];
    int base = e0;
#pragma unroll
    for (int q = 0; q < NC; q++) {
        start[q] = 0;
        if (q * 64 < n) {
            int incl = deg[q];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off)
                    incl += t;
            }
            start[q] = base + incl - deg[q];
            base += __shfl(incl, 63, 64);
            const int vl = q * 64 + lane;
            if (vl < n) {
                const int v = n0 + vl;
                row_ptr[v] = start[q];
                dinv[v] = 1.0f / sqrtf(1.0f + (float)deg[q]);
                const int dcl = deg[q] < 1 ? 1 : deg[q]; // gnn_builder_lib.h:1972-1982
                const float logd = logf((float)(dcl + 1));
                if (delta > 0.0f) { // (delta <= 0: the model has no PNA layer, the scalers are not needed)
                    amp[v] = logd / delta;
                    att[v] = delta / logd;
                }
                // default record: unused source slots alias the node itself
                int32_t *f = s_first[wave] + vl * 4;
                f[0] = v;
                f[1] = v;
                f[2] = v;
                f[3] = v;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- scatter: col[start[dst] + rank] = src, one store instruction per 64 edges
#pragma unroll
    for (int c = 0; c < EC; c++) {
        if (c < nec) { // wave-uniform: the cross-lane reads below run with every lane active
            const int d = ed[c] < 0 ? 0 : ed[c];
            int st = 0;
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const int t = __shfl(start[q], d & 63, 64);
                if ((d >> 6) == q)
                    st = t;
            }
            if (ed[c] >= 0) {
                col[st + erank[c]] = es[c];
                eid[st + erank[c]] = e0 + c * 64 + lane; // COO row of this CSR slot (gnn_builder_lib.h:1126-1166)
                if (erank[c] < 4)
                    s_first[wave][d * 4 + erank[c]] = es[c];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NC; q++) {
        const int vl = q * 64 + lane;
        if (vl < n) {
            const int32_t *f = s_first[wave] + vl * 4;
            node_rec[2 * (size_t)(n0 + vl)] = make_int4(start[q], deg[q], f[0], f[1]);
            node_rec[2 * (size_t)(n0 + vl) + 1] = make_int4(f[2], f[3], 0, 0);
        }
    }
    GNNB_STAMP_END(3);
    if (bad)
        flag_batch(err, err_host, 4);
}

hipError_t launch_graph_prep(const int32_t *coo, const int32_t *node_ptr, const int32_t *edge_ptr,
                             BatchTables &t, float pna_delta, int drop_self_loops, hipStream_t s)
{
    // t.err is zeroed when the workspace is created and again whenever it is read
    // (gnnb_workspace_check), so no per-batch memset node sits in front of this launch
    const int waves = t.num_graphs + 1;
    const int grid = (waves + (WG / 64) - 1) / (WG / 64);
    if (t.max_graph_nodes_hint > 0 && t.max_graph_nodes_hint <= 64)
        hipLaunchKernelGGL(k_graph_prep<64>, dim3(grid), dim3(WG), 0, s, (const int2 *)coo, node_ptr,
                           edge_ptr, t.num_graphs, t.num_nodes, t.num_edges, t.row_ptr, t.col, t.eid, t.node_rec,
                           t.dinv, t.amp, t.att, pna_delta, t.tile_first, t.tile_edge, t.tile_graph, t.graph_ptr, t.tile_rows,
                           t.num_tiles, t.max_graph_nodes_hint, drop_self_loops, t.err, t.err_host_dev);
    else
        hipLaunchKernelGGL(k_graph_prep<256>, dim3(grid), dim3(WG), 0, s, (const int2 *)coo, node_ptr,
                           edge_ptr, t.num_graphs, t.num_nodes, t.num_edges, t.row_ptr, t.col, t.eid, t.node_rec,
                           t.dinv, t.amp, t.att, pna_delta, t.tile_first, t.tile_edge, t.tile_graph, t.graph_ptr, t.tile_rows,
                           t.num_tiles, t.max_graph_nodes_hint, drop_self_loops, t.err, t.err_host_dev);
    return hipGetLastError();
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD, each XCD has
// its own L2).  Kernels whose neighbouring blocks touch neighbouring rows remap the block id so
// that each XCD owns one CONTIGUOUS run of chunks: a neighbour row fetched by the adjacent chunk
// is then an L2 hit instead of a second HBM fetch by another XCD.  Bijective for any grid size
// (cdna_hip_programming.md, "XCD swizzle must be bijective").  Speed only, never correctness.
__device__ inline int xcd_contiguous_block(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// =====================================================================================
// gather-aggregate
// =====================================================================================
// One workgroup owns a run of node tiles, i.e. a few WHOLE graphs (tiles are cut at graph
// boundaries), so every neighbour row a destination needs lies inside the workgroup's own
// node range.  The node rows are streamed from HBM exactly once, fully coalesced (16 B per
// lane), into LDS; the CSR slice (row_ptr, col) of the tile is staged next to them.  Each
// destination row is then reduced by a lane group (width/4 lanes, float4 per lane) reading
// its neighbours from LDS in CSR order -- no atomics, one owner per output row -- and the
// result is written back with 16-B coalesced stores.  Algorithmic HBM traffic per launch:
// 4*w*N read + 4*w*N*k_out write + 4*(N+1) + 4*E + 4*(T+1).
//
// A tile that does not fit the LDS budget (a graph far larger than tile_rows) takes the
// same code path with the neighbour rows read straight from global memory (L2).
//
// Semantics per mode:
//   GCN  gcn_conv_agg   gnn_builder_lib.h:1213-1289   sum_j x_j/sqrt((1+d_i)(1+d_j)) + x_i/sqrt((1+d_i)^2)
//   SUM  gin_conv_agg   gnn_builder_lib.h:1389-1437 + :1525-1535   sum_j x_j + x_i (1+eps)
//   MEAN sage_conv_agg  gnn_builder_lib.h:2161-2209   (sum_j x_j)/d, 0 when d = 0
//   PNA  pna_conv_agg   gnn_builder_lib.h:1750-1834 with h_ij = q_i + p_j (the per-edge
//        W_pre [x_i || x_j] + b split into two per-node products) and PyG's std:
//        sqrt(max(E[h^2]-E[h]^2, 1e-5)) zeroed where <= sqrt(1e-5)  (SURVEY finding 5).
// Neighbours are summed in CSR (= stable COO) order, the self term last, as the reference does.

template <int VEC>
struct Vf;
template <>
struct Vf<4> {
    float4 v;
    __device__ static Vf load(const float *p) { Vf r; r.v = *reinterpret_cast<const float4 *>(p); return r; }
    __device__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = v; }
    __device__ static Vf splat(float s) { Vf r; r.v = make_float4(s, s, s, s); return r; }
};
template <>
struct Vf<1> {
    float v;
    __device__ static Vf load(const float *p) { Vf r; r.v = *p; return r; }
    __device__ void store(float *p) const { *p = v; }
    __device__ static Vf splat(float s) { Vf r; r.v = s; return r; }
};
#define VF_BINOP(NAME, EXPR)                                                         \
    __device__ inline Vf<4> NAME(const Vf<4> &a, const Vf<4> &b)                     \
    {                                                                                \
        Vf<4> r;                                                                     \
        { const float x = a.v.x, y = b.v.x; r.v.x = (EXPR); }                        \
        { const float x = a.v.y, y = b.v.y; r.v.y = (EXPR); }                        \
        { const float x = a.v.z, y = b.v.z; r.v.z = (EXPR); }                        \
        { const float x = a.v.w, y = b.v.w; r.v.w = (EXPR); }                        \
        return r;                                                                    \
    }                                                                                \
    __device__ inline Vf<1> NAME(const Vf<1> &a, const Vf<1> &b)                     \
    {                                                                                \
        Vf<1> r;                                                                     \
        const float x = a.v, y = b.v;                                                \
        r.v = (EXPR);                                                                \
        return r;                                                                    \
    }
VF_BINOP(vadd, x + y)
VF_BINOP(vmul, x *y)
VF_BINOP(vmax, fmaxf(x, y))
VF_BINOP(vmin, fminf(x, y))
VF_BINOP(vdiv, x / y)
VF_BINOP(vsub, x - y)
#undef VF_BINOP
// PyG StdAggregation: var = E[h^2] - E[h]^2 ; std = sqrt(clamp(var, 1e-5)) ; 0 where <= sqrt(1e-5)
__device__ inline float pyg_std1(float mean2, float mean)
{
    float var = mean2 - mean * mean;
    var = var < 1e-5f ? 1e-5f : var;
    const float sd = sqrtf(var);
    return sd <= sqrtf(1e-5f) ? 0.0f : sd;
}
__device__ inline Vf<4> pyg_std(const Vf<4> &m2, const Vf<4> &m)
{
    Vf<4> r;
    r.v = make_float4(pyg_std1(m2.v.x, m.v.x), pyg_std1(m2.v.y, m.v.y), pyg_std1(m2.v.z, m.v.z),
                      pyg_std1(m2.v.w, m.v.w));
    return r;
}
__device__ inline Vf<1> pyg_std(const Vf<1> &m2, const Vf<1> &m)
{
    Vf<1> r;
    r.v = pyg_std1(m2.v, m.v);
    return r;
}

// -------------------------------------------------------------------------------------
// LDS-DMA helpers (global_load_lds: global -> LDS without VGPR staging).
typedef __attribute__((address_space(3))) void *lds_vptr;
typedef const __attribute__((address_space(1))) void *glb_vptr;

// LDS destination = wave-uniform base + lane * size (cdna_hip_programming.md section 5, Caveat);
// the size argument must be a literal, so one function per width.
__device__ inline void dma16_to_lds(const void *gsrc_lane, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((glb_vptr)gsrc_lane, (lds_vptr)lds_wave_base, 16, 0, 0);
}
__device__ inline void dma4_to_lds(const void *gsrc_lane, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((glb_vptr)gsrc_lane, (lds_vptr)lds_wave_base, 4, 0, 0);
}

// copy `count` dwords global -> LDS, spread over the workgroup's waves, 64 dwords per instruction
__device__ inline void dma_dwords(const void *g, void *l, int count, int wave, int lane, int nwaves)
{
    const char *gs = reinterpret_cast<const char *>(g);
    char *ls = reinterpret_cast<char *>(l);
    for (int c = wave * 64; c < count; c += nwaves * 64)
        if (c + lane < count)
            dma4_to_lds(gs + (size_t)(c + lane) * 4, ls + (size_t)c * 4);
}

// "Untracked" forms for software-pipelined kernels.  The compiler's waitcnt pass cannot tell which
// LDS bytes an in-flight LDS-DMA will write (dynamic shared memory carries no alias scopes), so after
// the builtin it puts s_waitcnt vmcnt(0) in front of EVERY later ds_read -- which serialises "issue
// the next stage's DMA, then compute on the current stage" completely.  Issued from inline assembly
// the DMA is invisible to that pass; the kernel then owns the ordering and MUST wait itself
// (dma_wait_all / a counted s_waitcnt, then a barrier) before any wave reads the destination.
// Compiler-inserted vmcnt waits stay correct: extra outstanding operations only make vmcnt(N) stronger.
__device__ inline void dma16_to_lds_u(const void *gsrc_lane, void *lds_wave_base)
{
    const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_vptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(a), "v"(gsrc_lane) : "memory");
}
// scalar-base form: address = wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset; the LDS
// destination is a wave-uniform LDS byte address.  Keeps a streaming kernel's per-chunk address arithmetic on the
// scalar unit (fp32 MFMA and VALU instructions share one issue port).
__device__ inline void dma16_to_lds_s(const void *gbase_uniform, uint32_t lane_byte_off, uint32_t lds_addr_uniform)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr_uniform), "v"(lane_byte_off),
                 "s"(gbase_uniform)
                 : "memory");
}
__device__ inline void dma4_to_lds_u(const void *gsrc_lane, void *lds_wave_base)
{
    const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_vptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(a), "v"(gsrc_lane) : "memory");
}
__device__ inline void dma_dwords_u(const void *g, void *l, int count, int wave, int lane, int nwaves)
{
    const char *gs = reinterpret_cast<const char *>(g);
    char *ls = reinterpret_cast<char *>(l);
    for (int c = wave * 64; c < count; c += nwaves * 64)
        if (c + lane < count)
            dma4_to_lds_u(gs + (size_t)(c + lane) * 4, ls + (size_t)c * 4);
}
__device__ inline void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n: the instruction takes an immediate
__device__ __forceinline__ void vmcnt_wait_upto(int n)
{
    switch (__builtin_amdgcn_readfirstlane(n)) { // scalar branch

#define GNNB_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    GNNB_VMW(1) GNNB_VMW(2) GNNB_VMW(3) GNNB_VMW(4) GNNB_VMW(5) GNNB_VMW(6) GNNB_VMW(7) GNNB_VMW(8) GNNB_VMW(9)
    GNNB_VMW(10) GNNB_VMW(11) GNNB_VMW(12) GNNB_VMW(13) GNNB_VMW(14) GNNB_VMW(15) GNNB_VMW(16) GNNB_VMW(17) GNNB_VMW(18)
#undef GNNB_VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}


// vmcnt wait with a run-time, wave-uniform count (the instruction takes an immediate): 0..63
__device__ __forceinline__ void vmcnt_wait_n(int n)
{
    switch (__builtin_amdgcn_readfirstlane(n)) { // scalar jump table
#define GNNB_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define GNNB_VMW8(b) GNNB_VMW(b) GNNB_VMW(b + 1) GNNB_VMW(b + 2) GNNB_VMW(b + 3) GNNB_VMW(b + 4) GNNB_VMW(b + 5) GNNB_VMW(b + 6) GNNB_VMW(b + 7)
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    GNNB_VMW(1) GNNB_VMW(2) GNNB_VMW(3) GNNB_VMW(4) GNNB_VMW(5) GNNB_VMW(6) GNNB_VMW(7)
    GNNB_VMW(8) GNNB_VMW(9) GNNB_VMW(10) GNNB_VMW(11) GNNB_VMW(12) GNNB_VMW(13) GNNB_VMW(14) GNNB_VMW(15)
    GNNB_VMW(16) GNNB_VMW(17) GNNB_VMW(18) GNNB_VMW(19) GNNB_VMW(20) GNNB_VMW(21) GNNB_VMW(22) GNNB_VMW(23)
    GNNB_VMW(24) GNNB_VMW(25) GNNB_VMW(26) GNNB_VMW(27) GNNB_VMW(28) GNNB_VMW(29) GNNB_VMW(30) GNNB_VMW(31)
    GNNB_VMW(32) GNNB_VMW(33) GNNB_VMW(34) GNNB_VMW(35) GNNB_VMW(36) GNNB_VMW(37) GNNB_VMW(38) GNNB_VMW(39)
    GNNB_VMW(40) GNNB_VMW(41) GNNB_VMW(42) GNNB_VMW(43) GNNB_VMW(44) GNNB_VMW(45) GNNB_VMW(46) GNNB_VMW(47)
    GNNB_VMW(48) GNNB_VMW(49) GNNB_VMW(50) GNNB_VMW(51) GNNB_VMW(52) GNNB_VMW(53) GNNB_VMW(54) GNNB_VMW(55)
    GNNB_VMW(56) GNNB_VMW(57) GNNB_VMW(58) GNNB_VMW(59) GNNB_VMW(60) GNNB_VMW(61) GNNB_VMW(62)
#undef GNNB_VMW8
#undef GNNB_VMW
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}

// 16-B (or 4-B) row-piece store, optionally non-temporal (the output is not re-read by this kernel)
typedef float agg_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void agg_store(const Vf<4> &v, float *p)
{
    if (NT) {
        agg_f32x4 t = {v.v.x, v.v.y, v.v.z, v.v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<agg_f32x4 *>(p));
    } else {
        v.store(p);
    }
}
template <bool NT>
__device__ __forceinline__ void agg_store(const Vf<1> &v, float *p)
{
    if (NT)
        __builtin_nontemporal_store(v.v, p);
    else
        v.store(p);
}

// LGConv normaliser (gnn_builder_lib.h:2383-2386): 1/sqrt(d_i d_j) on IN-degrees, 0 when the product is 0
__device__ __forceinline__ float lg_coef(int di, int dj)
{
    const int pr = di * dj;
    return pr > 0 ? __frsqrt_rn((float)pr) : 0.0f; // (v_rsq_f32: 1 ulp; the oracle's 1/sqrt differs by < 2e-7 relative)
}

// number of rows an output row has (PNA: max | min | mean | std)
template <int MODE>
struct AggOut {
    static constexpr int K = MODE == GNNB_AGG_PNA ? 4 : 1;
};

// The reduction itself, shared by the LDS-staged and the direct forms: `take` one neighbour row at a
// time in CSR order, `done` adds the self term / finalises and stores.
template <int MODE, int VEC, bool NT>
struct AggAcc {
    typedef Vf<VEC> V;
    V acc, vmx, vmn, s2;
    __device__ inline void init() { acc = V::splat(0.0f); vmx = acc; vmn = acc; s2 = acc; }
    // coef: GCN dinv_i * dinv_j, LG 1/sqrt(d_i d_j); xi: PNA's per-destination term q_i
    __device__ inline void take(const V &v, float coef, const V &xi, bool first)
    {
        if (MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_LG) {
            acc = vadd(acc, vmul(v, V::splat(coef)));
        } else if (MODE == GNNB_AGG_PNA) {
            const V h = vadd(xi, v);
            if (first) {
                vmx = h;
                vmn = h;
            } else {
                vmx = vmax(vmx, h);
                vmn = vmin(vmn, h);
            }
            acc = vadd(acc, h);
            s2 = vadd(s2, vmul(h, h));
        } else {
            acc = vadd(acc, v);
        }
    }
    __device__ inline void done(const V &xi, float di, int deg, float eps, float *__restrict__ out, size_t node, int w, int fo)
    {
        if (MODE == GNNB_AGG_GCN) {
            agg_store<NT>(vadd(acc, vmul(xi, V::splat(di * di))), out + node * w + fo);
        } else if (MODE == GNNB_AGG_SUM) {
            agg_store<NT>(vadd(acc, vmul(xi, V::splat(1.0f + eps))), out + node * w + fo);
        } else if (MODE == GNNB_AGG_MEAN) {
            // one reciprocal per row instead of a division per component (<= 1 ulp from sum / count)
            agg_store<NT>(deg > 0 ? vmul(acc, V::splat(1.0f / (float)deg)) : acc, out + node * w + fo);
        } else if (MODE == GNNB_AGG_PNA) {
            V mean = V::splat(0.0f), sd = V::splat(0.0f);
            if (deg > 0) {
                const V dn = V::splat((float)deg);
                mean = vdiv(acc, dn);
                sd = pyg_std(vdiv(s2, dn), mean);
            }
            float *o = out + node * 4 * w + fo;
            agg_store<NT>(vmx, o);
            agg_store<NT>(vmn, o + w);
            agg_store<NT>(mean, o + 2 * (size_t)w);
            agg_store<NT>(sd, o + 3 * (size_t)w);
        } else if (MODE == GNNB_AGG_COPY) {
            agg_store<NT>(xi, out + node * w + fo);
        } else { // LG, SIMPLE: no self term
            agg_store<NT>(acc, out + node * w + fo);
        }
    }
};

// One destination row reduced from an LDS stage that holds its whole graph: rows `sx`, node records
// `srec`, GCN normalisers `sdinv`, and (PNA, QLDS) the per-destination terms `sq`, all indexed by
// row - nb.  begin() issues every LDS read of the row (its record gives the first four sources, unused
// slots alias the row itself), finish() reduces and stores: two rows are begun before either is
// finished so that ten ds_read_b128 are in flight per lane.
template <int MODE, int VEC, bool QLDS = false, bool NT = false>
struct LdsRow {
    typedef Vf<VEC> V;
    int node, rp0, deg, jr[4];
    float di, sj[4];
    V xi, nbv[4];
    bool valid;

    __device__ inline void begin(bool ok, int nb, int r_, const float *sx, const float *sq, const int4 *srec,
                                 const float *sdinv, const float *__restrict__ selfq, int w, int fo)
    {
        valid = ok;
        if (!ok)
            return;
        node = nb + r_;
        if (MODE == GNNB_AGG_COPY) {
            xi = V::load(sx + (size_t)r_ * w + fo);
            return;
        }
        const int4 r0 = srec[2 * r_], r1 = srec[2 * r_ + 1];
        rp0 = r0.x;
        deg = r0.y;
        jr[0] = r0.z - nb;
        jr[1] = r0.w - nb;
        jr[2] = r1.x - nb;
        jr[3] = r1.y - nb;
        if (MODE == GNNB_AGG_PNA)
            xi = QLDS ? V::load(sq + (size_t)r_ * w + fo) : V::load(selfq + (size_t)node * w + fo);
        else if (MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_SUM)
            xi = V::load(sx + (size_t)r_ * w + fo);
#pragma unroll
        for (int q = 0; q < 4; q++)
            nbv[q] = V::load(sx + (size_t)jr[q] * w + fo); // unused slots alias the row itself
        if (MODE == GNNB_AGG_GCN) {
            di = sdinv[r_];
#pragma unroll
            for (int q = 0; q < 4; q++)
                sj[q] = di * sdinv[jr[q]];
        } else if (MODE == GNNB_AGG_LG) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                sj[q] = lg_coef(deg, srec[2 * jr[q]].y);
        }
    }

    __device__ inline void finish(int nb, const float *sx, const int4 *srec, const float *sdinv,
                                  const int32_t *__restrict__ col, float *__restrict__ out, int w, int fo,
                                  float eps)
    {
        if (!valid)
            return;
        AggAcc<MODE, VEC, NT> a;
        a.init();
        if (MODE != GNNB_AGG_COPY) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (deg > q)
                    a.take(nbv[q], sj[q], xi, q == 0);
            for (int k = rp0 + 4; k < rp0 + deg; k++) { // degree > 4: the rest of the CSR row
                const int j = col[k] - nb;
                const V v = V::load(sx + (size_t)j * w + fo);
                float c = 0.0f;
                if (MODE == GNNB_AGG_GCN)
                    c = di * sdinv[j];
                else if (MODE == GNNB_AGG_LG)
                    c = lg_coef(deg, srec[2 * j].y);
                a.take(v, c, xi, false);
            }
        }
        a.done(xi, di, deg, eps, out, (size_t)node, w, fo);
    }
};

// The same row straight from global memory (a graph larger than an LDS stage): neighbour rows are
// L2-side gathers, the degree comes from the node record (row_ptr holds row STARTS only: dropped edges
// leave gaps at the end of a graph's CSR segment, so row_ptr[v+1] - row_ptr[v] is not a degree).
template <int MODE, int VEC, bool NT>
__device__ inline void agg_row_direct(int node, const float *__restrict__ x, const float *__restrict__ selfq,
                                      float *__restrict__ out, const int4 *__restrict__ node_rec,
                                      const int32_t *__restrict__ col, const float *__restrict__ dinv, int w, int fo,
                                      float eps)
{
    typedef Vf<VEC> V;
    AggAcc<MODE, VEC, NT> a;
    a.init();
    int rp0 = 0, deg = 0;
    if (MODE != GNNB_AGG_COPY) {
        const int4 r0 = node_rec[2 * (size_t)node];
        rp0 = r0.x;
        deg = r0.y;
    }
    const float di = (MODE == GNNB_AGG_GCN) ? dinv[node] : 0.0f;
    V xi = V::splat(0.0f);
    if (MODE == GNNB_AGG_PNA)
        xi = V::load(selfq + (size_t)node * w + fo);
    else if (MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_SUM || MODE == GNNB_AGG_COPY)
        xi = V::load(x + (size_t)node * w + fo);
    for (int k = rp0; k < rp0 + deg; k++) {
        const int j = col[k];
        const V xj = V::load(x + (size_t)j * w + fo);
        float c = 0.0f;
        if (MODE == GNNB_AGG_GCN)
            c = di * dinv[j];
        else if (MODE == GNNB_AGG_LG)
            c = lg_coef(deg, node_rec[2 * (size_t)j].y);
        a.take(xj, c, xi, k == rp0);
    }
    a.done(xi, di, deg, eps, out, (size_t)node, w, fo);
}

// -------------------------------------------------------------------------------------
// Ring form: persistent, ONE workgroup of up to 16 waves per CU, whose waves share a ring of `nslots` big LDS stages
// (two stages take the whole 160 KB: ~140 rows each at w = 128, so molecules and graphs of a few hundred nodes fit).
// The workgroup walks a contiguous run of node tiles (whole graphs).  Per stage every wave fires its share of the
// LDS-DMA of the rows, node records, normalisers and the stage's CSR slice (global_load_lds, no VGPRs); stages retire
// in order behind a COUNTED vmcnt wait (VM operations retire in issue order, so "at most n younger operations
// outstanding" proves the stage has landed while the next stage's DMA and the previous stage's output stores stay in
// flight) and two barriers.  At the BASELINE sizes a CU's whole share of the input fits its ring, so all reads are in
// flight from the first microsecond; bigger batches cycle the ring.  A tile that does not fit a stage (one very large
// graph) is reduced straight from global memory.  Forms measured and dropped this round: a short-lived
// workgroup per tile group (round 1's default), one ring per wave.
static constexpr int RING_MAX_SLOTS = 4;

// Diagnostic build: wave 0 of every workgroup logs wall-clock stamps of its stage events (100 MHz ticks)
#ifdef GNNB_PROBE
#define RING_EV(code)                                                                                        \
    do {                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 2048 && pev < 31) {                                             \
            g_probe[blockIdx.x * 64 + 2 + 2 * pev] = wall_clock64();                                         \
            g_probe[blockIdx.x * 64 + 3 + 2 * pev] = (unsigned long long)(code);                             \
            pev++;                                                                                           \
            g_probe[blockIdx.x * 64 + 1] = pev;                                                              \
        }                                                                                                    \
    } while (0)
#else
#define RING_EV(code) do { } while (0)
#endif

template <int MODE, int VEC, bool NT>
__global__ __launch_bounds__(1024) void k_aggregate_ring(
    const float *__restrict__ x, const float *__restrict__ selfq, float *__restrict__ out,
    const int4 *__restrict__ node_rec, const int32_t *__restrict__ col, const float *__restrict__ dinv,
    const int32_t *__restrict__ tile_first, const int32_t *__restrict__ tile_edge, int num_tiles, int N, int E, int w,
    int glog2, int cap, int ecap, int nslots, int slot_bytes, int slack, float eps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool HASQ = MODE == GNNB_AGG_PNA, HASREC = MODE != GNNB_AGG_COPY, HASDINV = MODE == GNNB_AGG_GCN;
    constexpr int KOUT = AggOut<MODE>::K;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    // the workgroup's waves share the ring: wave sw of sn issues 1/sn of a stage's DMA and reduces 1/sn of its rows
    const int sw = wave, sn = nw;
    const int t0 = (int)(((long long)blockIdx.x * num_tiles) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * num_tiles) / gridDim.x);
    if (t0 >= t1)
        return; // (workgroup-uniform)
#ifdef GNNB_PROBE
    int pev = 0;
    if (threadIdx.x == 0 && blockIdx.x < 2048)
        g_probe[blockIdx.x * 64] = wall_clock64();
#endif
    char *wbase = smem;
    const int off_q = cap * w * 4;
    const int off_rec = off_q + (HASQ ? cap * w * 4 : 0);
    const int off_dinv = off_rec + cap * 32;
    const int off_col = off_dinv + cap * 4; // the stage's CSR slice (rows of degree > 4 read it): ecap entries
    const int nvec = w / VEC;
    const int G = 1 << glog2;       // lanes per destination row
    const int groups = 64 >> glog2; // rows a wave reduces at once
    const int grp = lane >> glog2;
    const int gl = lane & (G - 1);
    const int fiter = (nvec + G - 1) / G;

    // window of the tile table in a register: lane l holds tile_first[wb + l]
    // (clamped to N: the table of a malformed batch may hold stale entries; a flagged batch must still stay in range)
    int wb = t0;
    int tfv = min(tile_first[min(wb + lane, num_tiles)], N);
    int tev = HASREC ? min(tile_edge[min(wb + lane, num_tiles)], E) : 0;
    const int tf_end = min(tile_first[t1], N); // end of this ring's node range
    int ts = t0; // next tile to plan

    int f_nb[RING_MAX_SLOTS], f_rows[RING_MAX_SLOTS], f_mark[RING_MAX_SLOTS], f_e0[RING_MAX_SLOTS];
    int nfifo = 0, vm = 0, issue_slot = 0, head_slot = 0;

    auto issue = [&](int slot, int nb_, int rows_, int e0_, int ne_) -> int {
        char *sb = wbase + (size_t)slot * slot_bytes;
        int ops = 0;
        const int bytes = rows_ * w * 4;
        {
            const char *gx = reinterpret_cast<const char *>(x + (size_t)nb_ * w);
            if (VEC == 4) {
                for (int c = sw * 1024; c < bytes; c += sn * 1024, ops++)
                    if (c + lane * 16 < bytes)
                        dma16_to_lds_u(gx + c + lane * 16, sb + c);
            } else {
                for (int c = sw * 64; c < rows_ * w; c += sn * 64, ops++)
                    if (c + lane < rows_ * w)
                        dma4_to_lds_u(gx + (size_t)(c + lane) * 4, sb + (size_t)c * 4);
            }
        }
        if (HASQ) {
            const char *gq = reinterpret_cast<const char *>(selfq + (size_t)nb_ * w);
            if (VEC == 4) {
                for (int c = sw * 1024; c < bytes; c += sn * 1024, ops++)
                    if (c + lane * 16 < bytes)
                        dma16_to_lds_u(gq + c + lane * 16, sb + off_q + c);
            } else {
                for (int c = sw * 64; c < rows_ * w; c += sn * 64, ops++)
                    if (c + lane < rows_ * w)
                        dma4_to_lds_u(gq + (size_t)(c + lane) * 4, sb + off_q + (size_t)c * 4);
            }
        }
        if (HASREC) {
            const char *grec = reinterpret_cast<const char *>(node_rec + 2 * (size_t)nb_);
            const int rbytes = rows_ * 32;
            for (int c = sw * 1024; c < rbytes; c += sn * 1024, ops++)
                if (c + lane * 16 < rbytes)
                    dma16_to_lds_u(grec + c + lane * 16, sb + off_rec + c);
        }
        if (HASDINV) {
            for (int c = sw * 64; c < rows_; c += sn * 64, ops++)
                if (c + lane < rows_)
                    dma4_to_lds_u(dinv + nb_ + c + lane, sb + off_dinv + (size_t)c * 4);
        }
        if (HASREC) { // (a tracked global read of col inside the reduction would drain this wave's whole pipeline)
            for (int c = sw * 64; c < ne_; c += sn * 64, ops++)
                if (c + lane < ne_)
                    dma4_to_lds_u(col + e0_ + c + lane, sb + off_col + (size_t)c * 4);
        }
        return ops;
    };

    // reduce one landed stage; returns the number of store instructions the wave issued
    auto compute = [&](int slot, int nb_, int rows_, int e0_) -> int {
        const char *sb = wbase + (size_t)slot * slot_bytes;
        const float *sx = reinterpret_cast<const float *>(sb);
        const float *sq = reinterpret_cast<const float *>(sb + off_q);
        const int4 *srec = reinterpret_cast<const int4 *>(sb + off_rec);
        const float *sdinv = reinterpret_cast<const float *>(sb + off_dinv);
        const int32_t *scol = reinterpret_cast<const int32_t *>(sb + off_col) - e0_; // indexed by the CSR slot itself
        int nst = 0;
        for (int rb = sw * 2 * groups; rb < rows_; rb += sn * 2 * groups) {
            const bool has_b = rb + groups < rows_; // wave-uniform: the second row's store exists or not for the whole wave
            for (int f = gl; f < nvec; f += G) {
                const int fo = f * VEC;
                LdsRow<MODE, VEC, true, NT> A, B;
                A.begin(rb + grp < rows_, nb_, rb + grp, sx, sq, srec, sdinv, selfq, w, fo);
                if (has_b)
                    B.begin(rb + groups + grp < rows_, nb_, rb + groups + grp, sx, sq, srec, sdinv, selfq, w, fo);
                A.finish(nb_, sx, srec, sdinv, scol, out, w, fo, eps);
                if (has_b)
                    B.finish(nb_, sx, srec, sdinv, scol, out, w, fo, eps);
            }
            nst += KOUT * fiter * (has_b ? 2 : 1);
        }
        return nst;
    };

    for (;;) {
        // ---- fill the ring: plan greedy stages of whole tiles and fire their DMA
        while (nfifo < nslots && ts < t1) {
            int rel = __builtin_amdgcn_readfirstlane(ts - wb);
            if (rel >= 32 && wb + 63 < t1) { // slide the window (a tracked load: drain first so the counts stay exact)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wb = ts;
                tfv = min(tile_first[min(wb + lane, num_tiles)], N);
                tev = HASREC ? min(tile_edge[min(wb + lane, num_tiles)], E) : 0;
                rel = 0;
            }
            const int nb_ = __builtin_amdgcn_readlane(tfv, rel);
            const int e0_ = __builtin_amdgcn_readlane(tev, rel);
            // a stage = a run of whole tiles whose rows AND CSR slice fit a slot.  The remaining rows are cut into
            // EQUAL stages (a greedy cut leaves a tiny last stage, and every stage costs a memory latency when the
            // ring is shallower than the range): aim at remaining / ceil(remaining / cap) rows, + half a tile
            const int rem = max(tf_end - nb_, 1);
            const int nrem = (rem + cap - 1) / cap;
            const int cap_eff = min(cap, (rem + nrem - 1) / nrem + slack);
            unsigned long long m = __ballot(lane > rel && wb + lane <= t1 && tfv - nb_ <= cap_eff && tev - e0_ <= ecap &&
                                            tev >= e0_);
            if (m == 0 && cap_eff < cap)
                m = __ballot(lane > rel && wb + lane <= t1 && tfv - nb_ <= cap && tev - e0_ <= ecap && tev >= e0_);
            int te;
            bool big = false;
            if (m == 0) { // the next tile alone exceeds a stage
                te = ts + 1;
                big = true;
            } else {
                te = wb + 63 - __builtin_clzll(m);
            }
            const int rows_ = __builtin_amdgcn_readlane(tfv, __builtin_amdgcn_readfirstlane(te - wb)) - nb_;
            const int ne_ = big ? 0 : __builtin_amdgcn_readlane(tev, __builtin_amdgcn_readfirstlane(te - wb)) - e0_;
            ts = te;
            if (rows_ <= 0)
                continue;
            if (big) {
                for (int r = sw * groups + grp; r < rows_; r += sn * groups)
                    for (int f = gl; f < nvec; f += G)
                        agg_row_direct<MODE, VEC, NT>(nb_ + r, x, selfq, out, node_rec, col, dinv, w, f * VEC, eps);
                continue;
            }
            vm += issue(issue_slot, nb_, rows_, e0_, ne_);
            RING_EV(1000000 + rows_); // stage issued
#pragma unroll
            for (int i = 0; i < RING_MAX_SLOTS; i++)
                if (i == nfifo) {
                    f_nb[i] = nb_;
                    f_rows[i] = rows_;
                    f_mark[i] = vm;
                    f_e0[i] = e0_;
                }
            nfifo++;
            issue_slot = issue_slot + 1 == nslots ? 0 : issue_slot + 1;
        }
        if (nfifo == 0)
            break;
        // ---- retire the oldest stage: everything issued after its DMA may stay in flight
        RING_EV(2000000 + f_rows[0]); // waiting for the oldest stage
        vmcnt_wait_n(min(vm - f_mark[0], 63));
        asm volatile("s_barrier" ::: "memory"); // every wave's share of the stage has landed
        RING_EV(3000000 + f_rows[0]); // landed
        vm += compute(head_slot, f_nb[0], f_rows[0], f_e0[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // its LDS reads are done before the slot is refilled
        asm volatile("s_barrier" ::: "memory"); // ... by every wave
        RING_EV(4000000 + f_rows[0]); // reduced, stores issued
#pragma unroll
        for (int i = 0; i + 1 < RING_MAX_SLOTS; i++) {
            f_nb[i] = f_nb[i + 1];
            f_rows[i] = f_rows[i + 1];
            f_mark[i] = f_mark[i + 1];
            f_e0[i] = f_e0[i + 1];
        }
        nfifo--;
        head_slot = head_slot + 1 == nslots ? 0 : head_slot + 1;
    }
}

template <int MODE, int VEC>
static hipError_t launch_aggregate_ring_t(const BatchTables &t, const float *x, const float *selfq,
                                          float *out, int w, float eps, hipStream_t s)
{
    const Options &o = options();
    if (t.num_tiles <= 0)
        return hipSuccess;
    const int nvec = w / VEC;
    int glog2 = 0;
    while ((1 << glog2) < nvec && glog2 < 6)
        glog2++;
    static int num_cus = 0;
    if (num_cus == 0) {
        int devid = 0;
        hipDeviceProp_t prop;
        num_cus = (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess)
                      ? prop.multiProcessorCount : 256;
    }
    // per staged row: the row itself (PNA: p and q), its 32-B record, its normaliser, and 4 CSR entries (a stage
    // whose CSR slice is longer than 4 per row -- multigraphs, hubs -- is cut shorter by the planner)
    constexpr int ECAP_PER_ROW = 4;
    const size_t per_row = (size_t)w * 4 * (MODE == GNNB_AGG_PNA ? 2 : 1) + (MODE != GNNB_AGG_COPY ? 32 + 4 + 4 * ECAP_PER_ROW : 0);
    const int wgs = std::max(o.agg_ring_wg_per_cu, 1);
    const size_t budget = (size_t)(o.agg_lds_kb > 0 ? std::min(std::max(o.agg_lds_kb, 8), 158) : 158 / wgs) * 1024;
    int ns = std::min(std::max(o.agg_ring_slots, 1), RING_MAX_SLOTS);
    int nw = o.agg_ring_waves;
    // one ring per workgroup: stages as large as the budget allows
    if (nw <= 0)
        nw = 16; // (measured: 16 waves issue a stage's DMA and drain its stores faster than 8)
    nw = std::min(std::max(nw, 1), 16);
    int cap = (int)((budget / ns) / per_row);
    cap = std::min(std::max(cap, 1), 4096);
    const int slot_bytes = (int)((((size_t)cap * per_row) + 15) & ~(size_t)15);
    const size_t lds = (size_t)ns * slot_bytes;
    // persistent: `wgs` workgroups per CU; fewer when the batch has fewer tiles than rings
    int grid = num_cus * wgs;
    grid = std::min(grid, t.num_tiles);
    if (grid < 1)
        grid = 1;
    auto launch = [&](auto kern) -> hipError_t {
        {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
            if (e != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, s, x, selfq, out, t.node_rec, t.col, t.dinv,
                           t.tile_first, t.tile_edge, t.num_tiles, t.num_nodes, t.num_edges, w, glog2, cap,
                           cap * ECAP_PER_ROW, ns, slot_bytes, std::max(t.tile_rows / 2, 1) + 2, eps);
        return hipGetLastError();
    };
    if (o.agg_nt_store)
        return launch(k_aggregate_ring<MODE, VEC, true>);
    return launch(k_aggregate_ring<MODE, VEC, false>);
}

// -------------------------------------------------------------------------------------
// GINE aggregate (reference gine_conv_agg + the self term of gine_conv, gnn_builder_lib.h:1555-1742):
//   out_i = (1 + eps) x_i + sum_{j -> i} relu(x_j + p_e),   p_e = W_e e_ij + b_e  (projected by the GEMM kernel,
// [E, w] rows in COO order; the CSR slot's COO row comes from the edge-index table graph prep writes,
// compute_neighbor_and_edge_index_tables :1126-1166).  One lane group per destination row, rows and edge terms
// gathered straight from L2: GINE is not on a BASELINE workload, this is the plain form.
template <int VEC>
__global__ __launch_bounds__(WG) void k_aggregate_edges(const float *__restrict__ x, const float *__restrict__ eterm,
                                                        float *__restrict__ out, const int4 *__restrict__ node_rec,
                                                        const int32_t *__restrict__ col, const int32_t *__restrict__ eid,
                                                        int N, int w, int glog2, float eps)
{
    typedef Vf<VEC> V;
    const int G = 1 << glog2, groups = WG >> glog2;
    const int grp = threadIdx.x >> glog2, gl = threadIdx.x & (G - 1);
    const int node = blockIdx.x * groups + grp;
    if (node >= N)
        return;
    const int4 r0 = node_rec[2 * (size_t)node];
    const int rp0 = r0.x, deg = r0.y;
    const int nvec = w / VEC;
    for (int f = gl; f < nvec; f += G) {
        const int fo = f * VEC;
        V acc = V::splat(0.0f);
        for (int k = rp0; k < rp0 + deg; k++) {
            const V xj = V::load(x + (size_t)col[k] * w + fo);
            const V pe = V::load(eterm + (size_t)eid[k] * w + fo);
            acc = vadd(acc, vmax(vadd(xj, pe), V::splat(0.0f))); // merge_sum_1d, activation_relu, sum_incremental
        }
        const V xi = V::load(x + (size_t)node * w + fo);
        vadd(acc, vmul(xi, V::splat(1.0f + eps))).store(out + (size_t)node * w + fo);
    }
}

hipError_t launch_aggregate_edges(const BatchTables &t, const float *x, const float *eterm, float *out, int width,
                                  float eps, hipStream_t s)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    const bool v4 = (width % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)out & 15) == 0) &&
                    (((uintptr_t)eterm & 15) == 0);
    const int nvec = v4 ? width / 4 : width;
    int glog2 = 0;
    while ((1 << glog2) < nvec && glog2 < 6)
        glog2++;
    const int groups = WG >> glog2;
    const int grid = (t.num_nodes + groups - 1) / groups;
    if (v4)
        hipLaunchKernelGGL(k_aggregate_edges<4>, dim3(grid), dim3(WG), 0, s, x, eterm, out, t.node_rec, t.col, t.eid,
                           t.num_nodes, width, glog2, eps);
    else
        hipLaunchKernelGGL(k_aggregate_edges<1>, dim3(grid), dim3(WG), 0, s, x, eterm, out, t.node_rec, t.col, t.eid,
                           t.num_nodes, width, glog2, eps);
    return hipGetLastError();
}

hipError_t launch_aggregate(const BatchTables &t, int kind, const float *x, const float *selfq,
                            float *out, int width, float eps, hipStream_t s)
{
    const bool v4 = (width % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)out & 15) == 0) &&
                    (selfq == nullptr || ((uintptr_t)selfq & 15) == 0);
#define GNNB_AGG_CASE(K)                                                                         \
    case K:                                                                                      \
        return v4 ? launch_aggregate_ring_t<K, 4>(t, x, selfq, out, width, eps, s)               \
                  : launch_aggregate_ring_t<K, 1>(t, x, selfq, out, width, eps, s);
    switch (kind) {
        GNNB_AGG_CASE(GNNB_AGG_GCN)
        GNNB_AGG_CASE(GNNB_AGG_SUM)
        GNNB_AGG_CASE(GNNB_AGG_MEAN)
        GNNB_AGG_CASE(GNNB_AGG_PNA)
        GNNB_AGG_CASE(GNNB_AGG_LG)
        GNNB_AGG_CASE(GNNB_AGG_SIMPLE)
        GNNB_AGG_CASE(GNNB_AGG_COPY)
    default:
        return hipErrorInvalidValue;
    }
#undef GNNB_AGG_CASE
}

// =====================================================================================
// dense update: multi-segment  Y = act( sum_s (rs_s . A_s) W_s^T + bias + skip )
// =====================================================================================
// Reference: `linear` applied to one node vector at a time (gnn_builder_lib.h:808-905) inside
// every conv (gcn :1379, gin :1538-1544, sage, pna :2146-2147) and the MLP head
// (templates/model.cpp.jinja:454-530).  Here all M rows of the batch go through one GEMM on
// the fp32 matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation; gfx950
// has no xf32).  Both operands are K-contiguous ("NT" GEMM: activations [M,K] row-major,
// weights [N,K] row-major = torch Linear layout), so A and W tiles are staged identically:
// 16-B global loads -> registers -> LDS rows padded to 36 floats (conflict-free
// ds_read_b128).  One ds_read_b128 per operand feeds four MFMA k-steps: lane (i, h) holds
// k = kb+4h..kb+4h+3, and MFMA step s contracts k in {kb+s, kb+4+s} -- a permutation of the
// k order shared by A and W, which the sum does not care about.
// Segments let SAGE ([mean | x] . [Wl | Wr]^T) and PNA ([x | A | amp.A | att.A] . Wpost^T, 13F
// wide) run as ONE GEMM without materialising the concatenation in HBM: the per-row scaler is
// applied while the A tile is staged.
static constexpr int BM = 128;
static constexpr int BK = 32;
static constexpr int LDS_LD = BK + 4; // padded row, floats

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline float act_apply(float v, int act)
{
    switch (act) {
    case GNNB_ACT_RELU:
        return v > 0.0f ? v : 0.0f; // gnn_builder_lib.h:363-375
    case GNNB_ACT_GELU:
        return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); // nn.GELU (erf), lib:378-385
    case GNNB_ACT_SIGMOID:
        return 1.0f / (1.0f + expf(-v)); // lib:420-425
    case GNNB_ACT_TANH:
        return tanhf(v); // lib:436-448
    default:
        return v;
    }
}

// Compile-time activation: the epilogues dispatch on `act` ONCE and run a straight-line copy of
// the store loop per activation (a runtime switch inside the unrolled loops replicates the inlined
// erff/tanhf/expf bodies per element: thousands of instructions and hundreds of branches).
template <int ACT>
__device__ inline float act_t(float v)
{
    if (ACT == GNNB_ACT_RELU)
        return v > 0.0f ? v : 0.0f;
    if (ACT == GNNB_ACT_GELU)
        return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (ACT == GNNB_ACT_SIGMOID)
        return 1.0f / (1.0f + expf(-v));
    if (ACT == GNNB_ACT_TANH)
        return tanhf(v);
    return v;
}
// calls f(IntTag<act>{}) with `act` turned into a compile-time constant
#define GNNB_DISPATCH_ACT(act, f)                        \
    switch (act) {                                       \
    case GNNB_ACT_RELU: f(IntTag<GNNB_ACT_RELU>{}); break;       \
    case GNNB_ACT_GELU: f(IntTag<GNNB_ACT_GELU>{}); break;       \
    case GNNB_ACT_SIGMOID: f(IntTag<GNNB_ACT_SIGMOID>{}); break; \
    case GNNB_ACT_TANH: f(IntTag<GNNB_ACT_TANH>{}); break;       \
    default: f(IntTag<GNNB_ACT_NONE>{}); break;                  \
    }

__device__ inline float4 load4_guard(const float *p, int remaining, bool vec)
{
    // `remaining` = number of valid floats at p (<= 0: none)
    if (remaining <= 0)
        return make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && remaining >= 4)
        return *reinterpret_cast<const float4 *>(p);
    float4 r;
    r.x = p[0];
    r.y = remaining > 1 ? p[1] : 0.f;
    r.z = remaining > 2 ? p[2] : 0.f;
    r.w = remaining > 3 ? p[3] : 0.f;
    return r;
}

template <int NT> // workgroup tile = 128 x (64*NT); wave tile = 64 x (32*NT)
__global__ __launch_bounds__(WG) void k_linear(GemmArgs g, const float *__restrict__ W, int ldw,
                                               const float *__restrict__ bias,
                                               const float *__restrict__ skip,
                                               float *__restrict__ Y, int M, int N, int act)
{
    constexpr int BN = 64 * NT;
    constexpr int BROWS = BN / 32; // W-tile staging passes per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);         // [2][BM*LDS_LD]
    float *Bs = As + 2 * BM * LDS_LD;                    // [2][BN*LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const int c4 = tid & 7;  // which float4 of the 32-wide k chunk
    const int r0 = tid >> 3; // 0..31

    f32x16 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < NT; ni++)
#pragma unroll
            for (int i = 0; i < 16; i++)
                acc[mi][ni][i] = 0.0f;

    float4 ra[4], rb[BROWS];
    const int total = g.cpre[g.nseg];

    auto load_chunk = [&](int c) {
        // segment lookup with static indexing only (keeps the kernarg struct out of scratch)
        const float *ap = g.a[0];
        const float *rs = g.rs[0];
        int lda = g.lda[0], ks = g.k[0], koff = g.koff[0], cbase = 0, av = g.avec[0], wv = g.wvec[0];
#pragma unroll
        for (int s = 1; s < 4; s++) {
            if (s < g.nseg && c >= g.cpre[s]) {
                ap = g.a[s];
                rs = g.rs[s];
                lda = g.lda[s];
                ks = g.k[s];
                koff = g.koff[s];
                cbase = g.cpre[s];
                av = g.avec[s];
                wv = g.wvec[s];
            }
        }
        const int kk = (c - cbase) * BK + c4 * 4;
        const int rem = ks - kk;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int row = m0 + r0 + 32 * p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) {
                v = load4_guard(ap + (size_t)row * lda + kk, rem, av != 0);
                if (rs != nullptr) {
                    const float sc = rs[row];
                    v.x *= sc;
                    v.y *= sc;
                    v.z *= sc;
                    v.w *= sc;
                }
            }
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < BROWS; p++) {
            const int n = n0 + r0 + 32 * p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < N)
                v = load4_guard(W + (size_t)n * ldw + koff + kk, rem, wv != 0);
            rb[p] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        float *a = As + buf * BM * LDS_LD;
        float *b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int p = 0; p < 4; p++)
            *reinterpret_cast<float4 *>(a + (r0 + 32 * p) * LDS_LD + c4 * 4) = ra[p];
#pragma unroll
        for (int p = 0; p < BROWS; p++)
            *reinterpret_cast<float4 *>(b + (r0 + 32 * p) * LDS_LD + c4 * 4) = rb[p];
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int c = 0; c < total; c++) {
        const int buf = c & 1;
        if (c + 1 < total)
            load_chunk(c + 1); // global loads stay in flight under the MFMAs below
        const float *a = As + buf * BM * LDS_LD + (wm * 64 + li) * LDS_LD + 4 * lh;
        const float *b = Bs + buf * BN * LDS_LD + (wn * 32 * NT + li) * LDS_LD + 4 * lh;
#pragma unroll
        for (int kb = 0; kb < BK; kb += 8) {
            float4 fa[2], fb[NT];
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
                fa[mi] = *reinterpret_cast<const float4 *>(a + mi * 32 * LDS_LD + kb);
#pragma unroll
            for (int ni = 0; ni < NT; ni++)
                fb[ni] = *reinterpret_cast<const float4 *>(b + ni * 32 * LDS_LD + kb);
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int ni = 0; ni < NT; ni++) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].x, fb[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].y, fb[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].z, fb[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].w, fb[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        if (c + 1 < total)
            store_chunk(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    auto epilogue = [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int ni = 0; ni < NT; ni++) {
                const int colg = n0 + wn * 32 * NT + ni * 32 + li;
                if (colg >= N)
                    continue;
                const float bv = bias ? bias[colg] : 0.0f;
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const int rowg = m0 + wm * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    if (rowg < M) {
                        float v = acc[mi][ni][reg] + bv;
                        if (skip)
                            v += skip[(size_t)rowg * N + colg];
                        Y[(size_t)rowg * N + colg] = act_t<ACT>(v);
                    }
                }
            }
    };
    GNNB_DISPATCH_ACT(act, epilogue)
}


// ---- fp32 product through the bf16 matrix cores ("bf16x6").  x = h + m + l EXACTLY, each piece a bf16
// (8 significant bits each: truncate, subtract, truncate, subtract -- every step is exact in fp32), so
// a.b = sum of nine bf16 x bf16 products, each exact in fp32.  The six with i + j <= 2 are kept
// (hh, hm, mh, hl, lh, mm); the three dropped ones are below 2^-24 |a||b|, i.e. below what fp32 resolves of
// the product.  Accumulation is fp32 inside v_mfma_f32_16x16x32_bf16.  Cost: 6 MFMA of 4 passes per
// 32-wide k block instead of 8 fp32 MFMA of 8 passes -- 2.4x fewer pipe cycles, and fp32 MFMA runs at
// the vector-FMA rate on this chip (tools/micro/mfma_valu_overlap.hip).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3(float x, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xffff0000u;
    l = __float_as_uint(r1 - __uint_as_float(m)); // <= 8 significant bits left: its upper half is exact
}
// two fp32 bit patterns -> their upper halves packed as {bf16(a) in bits 0..15, bf16(b) in bits 16..31}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v)
{
    union {
        u32x4 u;
        bf16x8 b;
    } c;
    c.u = v;
    return c.b;
}

// split 8 consecutive fp32 values (two float4) into the three bf16x8 pieces of an MFMA operand
__device__ __forceinline__ void split3x8(const float4 &f0, const float4 &f1, u32x4 &h, u32x4 &m, u32x4 &l)
{
    const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t h0, m0, l0, h1, m1, l1;
        split3(v[2 * i], h0, m0, l0);
        split3(v[2 * i + 1], h1, m1, l1);
        h[i] = pack_hi16(h0, h1);
        m[i] = pack_hi16(m0, m1);
        l[i] = pack_hi16(l0, l1);
    }
}


// -------------------------------------------------------------------------------------
// LDS-DMA form of the tiled GEMM above for the regular case -- rows 16-B aligned, segment widths whole
// 32-wide chunks (GraphSAGE at d = 256: [mean | x] . [Wl | Wr]^T, K = 2 x 256; PNA at d = 128: 13 x 128 with
// two row-scaled segments, the scaler applied to the A fragments).
// Same 32x32x2 MFMA schedule (and summation order) as k_linear, but the A and W chunks go global -> LDS directly
// (untracked global_load_lds, no VGPR staging, no ds_write).  LDS rows are unpadded [row][32 floats]; 16-B pieces
// are XOR-swizzled through the DMA *source* address (slot = piece ^ (row & 7)), which keeps the ds_read_b128
// fragment reads conflict-free.
//
// Shape: two 4-wave workgroups per CU (they fill each other's barrier gaps), 128 x 128 output tile, two chunk buffers;
// the constants below also express the other shape that was built and measured -- ONE 8-wave workgroup per CU, 256 x
// 128 tile, three-deep chunk ring (DM 256, DWG 512, DNBUF 3, DWGPC 1): 579 / 544 us against 592 / 535 us at the C4 /
// C5 shapes, a wash, every barrier idles the whole CU.  What mattered was in the generated code: without its chunk DMA
// the kernel ran at 84 % of the fp32 MFMA peak, with it at 65 % -- see the note on compiler-tracked loads in the item
// body.
static constexpr int DM = 128, DN = 128, DWG = 256, DNBUF = 2, DNW = DWG / 64, DWGPC = 2;
static constexpr int DBUF_B = (DM + DN) * BK * 4; // 32 KB: A chunk | W chunk
// MATH 1 (opt-in, gnnb_set_option("math", 1)): the same chunks, but each 16-wide k block is multiplied as six
// v_mfma_f32_32x32x16_bf16 products of an exact 3-way bf16 split of BOTH operands (see split3), the fragments split in
// the wave after the LDS read -- 24 MFMA of 8 passes instead of 32 of 16 per k block and accumulator quartet.
template <int MATH>
__global__ __launch_bounds__(DWG) void k_linear_dma(GemmArgs g, const float *__restrict__ W, int ldw,
                                                    const float *__restrict__ bias,
                                                    const float *__restrict__ skip, float *__restrict__ Y, int M,
                                                    int N, int act, int tiles_m, int tiles_n, int split_from, int split)
{
    constexpr int NT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1; // 2 x 2 waves: 64 rows x 64 columns each
    const int total = g.cpre[g.nseg];
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t smem_a = (uint32_t)(uintptr_t)(lds_vptr)smem;
    // DMA lane geometry: an instruction covers 8 rows x eight 16-B pieces; LDS slot p of row r holds piece p ^ (r & 7)
    const int drow = lane >> 3;
    const uint32_t dpiece_b = (uint32_t)(((lane & 7) ^ drow) << 4);

    // PERSISTENT over work items (grid = what is resident: two workgroups per CU); the chunk pipeline runs straight
    // across item boundaries.  TAIL SPLIT: tiles / CUs is rarely whole (PNA at C4: 1153 tiles on 256 CUs = 4.5 per CU,
    // paid as 5).  Tiles from `split_from` on -- the last, partial round -- are handed out as `split` (2 or 4) row
    // slices each, so that the round costs a half or a quarter tile.  A slice keeps the tile's MFMA order per output
    // element: 64 rows = one 32-row accumulator block per wave instead of two, 32 rows = the same on half of the waves.
    const int num_tiles = tiles_m * tiles_n;
    const int num_items = split_from + split * (num_tiles - split_from);
    if ((int)blockIdx.x >= num_items)
        return;
    auto decode = [&](int it, int &m0, int &n0, int &mrows) {
        const bool part = it >= split_from;
        const int j = it - split_from;
        const int t = part ? split_from + j / split : it;
        mrows = part ? DM / split : DM;
        m0 = (t / tiles_n) * DM + (part ? (j % split) * mrows : 0);
        n0 = (t % tiles_n) * DN;
    };

    // issue cursor: runs ahead of the multiply cursor, across item boundaries (its item's origin is decoded once per item:
    // the integer divisions are scalar instructions in front of every wave's next MFMA)
    int iss_item = blockIdx.x, iss_c = 0, iss_buf = 0;
    int iss_m0 = 0, iss_n0 = 0, iss_mrows = 0;
    decode(iss_item, iss_m0, iss_n0, iss_mrows);
    int vm = 0; // vector-memory instructions this wave has issued (DMA + epilogue stores): for the counted waits
    // The next chunk's DMA goes out in FOUR parts, one per k step of the chunk being multiplied (a burst of eight
    // instructions behind the barrier kept every wave of the workgroup off the matrix pipe at the same moment):
    // issue_begin() resolves the addresses on the scalar unit, issue_part(j) fires part j, issue_end() moves the cursor.
    struct IssueCtx {
        const float *ga, *gw;
        uint32_t la, lw, lda_b, ldw_b;
        int ra_max, rw_max, mrows;
        bool valid;
    };
    auto issue_begin = [&]() -> IssueCtx {
        IssueCtx ic;
        ic.valid = iss_item < num_items;
        if (!ic.valid)
            return ic;
        const int m0 = iss_m0, n0 = iss_n0;
        ic.mrows = iss_mrows;
        const int c = iss_c;
        // segment lookup with static indexing only (keeps the kernarg struct out of scratch)
        const float *ap = g.a[0];
        int lda = g.lda[0], koff = g.koff[0], cbase = 0;
#pragma unroll
        for (int sgm = 1; sgm < 4; sgm++) {
            if (sgm < g.nseg && c >= g.cpre[sgm]) {
                ap = g.a[sgm];
                lda = g.lda[sgm];
                koff = g.koff[sgm];
                cbase = g.cpre[sgm];
            }
        }
        const int kk = (c - cbase) * BK;
        // scalar bases (tile origin, clamped into the matrix) + per-lane 32-bit offsets: the address arithmetic stays
        // on the scalar unit
        const int m0c = min(m0, M - 1), n0c = min(n0, N - 1);
        ic.ga = ap + (size_t)m0c * lda + kk;
        ic.gw = W + (size_t)n0c * ldw + koff + kk;
        ic.ra_max = M - 1 - m0c, ic.rw_max = N - 1 - n0c; // rows past M / N re-read the last valid row (never stored)
        ic.la = smem_a + (uint32_t)iss_buf * DBUF_B, ic.lw = ic.la + DM * BK * 4;
        ic.lda_b = (uint32_t)lda * 4, ic.ldw_b = (uint32_t)ldw * 4;
        return ic;
    };
    constexpr int DPARTS = BK / 8, DA_PER = DM / 8 / DNW, DW_PER = DN / 8 / DNW; // A / W instructions per wave and chunk
    static_assert(DA_PER <= DPARTS && DW_PER <= DPARTS, "one A and one W instruction per part at most");
    auto issue_part = [&](const IssueCtx &ic, int i) {
        if (!ic.valid)
            return;
        if (i < DA_PER) {
            const int r0 = (wave * DA_PER + i) * 8;
            if (r0 < ic.mrows) {
                dma16_to_lds_s(ic.ga, (uint32_t)min(r0 + drow, ic.ra_max) * ic.lda_b + dpiece_b, ic.la + (uint32_t)r0 * 128);
                vm++;
            }
        }
        if (i < DW_PER) {
            const int r0 = (wave * DW_PER + i) * 8;
            dma16_to_lds_s(ic.gw, (uint32_t)min(r0 + drow, ic.rw_max) * ic.ldw_b + dpiece_b, ic.lw + (uint32_t)r0 * 128);
            vm++;
        }
    };
    auto issue_end = [&](const IssueCtx &ic) -> int { // returns vm after the chunk's DMA (its "mark"), -1 when there was none
        if (!ic.valid)
            return -1;
        iss_buf = iss_buf + 1 == DNBUF ? 0 : iss_buf + 1;
        if (++iss_c == total) {
            iss_c = 0;
            iss_item += gridDim.x;
            if (iss_item < num_items)
                decode(iss_item, iss_m0, iss_n0, iss_mrows);
        }
        return vm;
    };
    auto issue_next = [&]() -> int {
        const IssueCtx ic = issue_begin();
#pragma unroll
        for (int i = 0; i < DPARTS; i++)
            issue_part(ic, i);
        return issue_end(ic);
    };

    // marks of the chunks in flight (DNBUF - 1 of them): mk0 = the chunk multiplied next, mk1 = the one after it
    int mk0 = issue_next(), mk1 = DNBUF > 2 ? issue_next() : -1;
    int buf = 0;
    const bool vec = (N % 4 == 0) && (((uintptr_t)Y & 15) == 0) && (bias == nullptr || ((uintptr_t)bias & 15) == 0) &&
                     (skip == nullptr || ((uintptr_t)skip & 15) == 0);

    // one work item with MC 32-row accumulator blocks per wave (2 = whole tile, 1 = a slice, 0 = a wave that only
    // keeps the chunk pipeline going).  A compile-time MC: with a run-time block count the accumulators of the
    // conditional block leave the AGPRs at every loop header.
    auto run_item = [&](auto mtag, int m0, int n0, int rbase) {
        constexpr int MC = decltype(mtag)::value;
        f32x16 acc[MC > 0 ? MC : 1][NT];
#pragma unroll
        for (int mi = 0; mi < (MC > 0 ? MC : 1); mi++)
#pragma unroll
            for (int ni = 0; ni < NT; ni++)
#pragma unroll
                for (int i = 0; i < 16; i++)
                    acc[mi][ni][i] = 0.0f;

        // per-row scalers of the scaled segments (PNA: amp . A, att . A), fetched once per item for the lane's A rows
        float sc[4][MC > 0 ? MC : 1];
        if (MC > 0) {
#pragma unroll
            for (int sgm = 0; sgm < 4; sgm++)
#pragma unroll
                for (int mi = 0; mi < MC; mi++) {
                    const int row = min(m0 + rbase + mi * 32 + li, M - 1);
                    sc[sgm][mi] = (sgm < g.nseg && g.rs[sgm] != nullptr) ? g.rs[sgm][row] : 1.0f;
                }
            // these are loads the compiler tracks: left pending, their first use INSIDE the chunk loop is guarded by
            // s_waitcnt vmcnt(0) in every iteration -- which also waits for the chunk DMA just issued, i.e. serialises
            // "request the next chunk" and "multiply this one" (found in round 2: the kernel had been running that
            // way).  Consumed here, once per item; the chunk loop then has no tracked load in flight.
#pragma unroll
            for (int sgm = 0; sgm < 4; sgm++)
#pragma unroll
                for (int mi = 0; mi < MC; mi++)
                    asm volatile("" : "+v"(sc[sgm][mi]));
        }

        for (int c = 0; c < total; c++) {
            // this chunk has landed for this wave when at most the operations issued after it are outstanding (VM
            // operations retire in order; loads the compiler tracks itself only make the wait stricter) ...
            vmcnt_wait_n(min(vm - mk0, 63));
            __syncthreads(); // ... and for everyone; and everyone is done reading the buffer refilled next
            const IssueCtx ic = issue_begin();
            if (MC == 0) {
#pragma unroll
                for (int i = 0; i < DPARTS; i++)
                    issue_part(ic, i);
            }
            const float *a = reinterpret_cast<const float *>(smem + (size_t)buf * DBUF_B);
            const float *b = reinterpret_cast<const float *>(smem + (size_t)buf * DBUF_B + DM * BK * 4);
            buf = buf + 1 == DNBUF ? 0 : buf + 1;
            if (MC > 0) {
                float s[MC > 0 ? MC : 1]; // this chunk's segment (uniform), static indexing
                bool scaled = g.rs[0] != nullptr;
#pragma unroll
                for (int mi = 0; mi < MC; mi++)
                    s[mi] = sc[0][mi];
#pragma unroll
                for (int sgm = 1; sgm < 4; sgm++)
                    if (sgm < g.nseg && c >= g.cpre[sgm]) {
#pragma unroll
                        for (int mi = 0; mi < MC; mi++)
                            s[mi] = sc[sgm][mi];
                        scaled = g.rs[sgm] != nullptr;
                    }
                if (MATH) {
                    // lane (li, lh) of a 32x32x16 bf16 MFMA holds k = 8 lh .. + 7 of row / column li for both operands:
                    // two 16-B pieces per fragment and k block
#pragma unroll
                    for (int kb2 = 0; kb2 < BK / 16; kb2++) {
                        u32x4 ah[MC > 0 ? MC : 1], am[MC > 0 ? MC : 1], al[MC > 0 ? MC : 1], wh[NT], wm[NT], wl[NT];
                        const int piece = 4 * kb2 + 2 * lh;
#pragma unroll
                        for (int mi = 0; mi < MC; mi++) {
                            const int r = rbase + mi * 32 + li;
                            float4 f0 = *reinterpret_cast<const float4 *>(a + r * BK + ((piece ^ (r & 7)) << 2));
                            float4 f1 = *reinterpret_cast<const float4 *>(a + r * BK + (((piece + 1) ^ (r & 7)) << 2));
                            if (scaled) {
                                f0.x *= s[mi], f0.y *= s[mi], f0.z *= s[mi], f0.w *= s[mi];
                                f1.x *= s[mi], f1.y *= s[mi], f1.z *= s[mi], f1.w *= s[mi];
                            }
                            split3x8(f0, f1, ah[mi], am[mi], al[mi]);
                        }
#pragma unroll
                        for (int ni = 0; ni < NT; ni++) {
                            const int r = wn * 32 * NT + ni * 32 + li;
                            const float4 f0 = *reinterpret_cast<const float4 *>(b + r * BK + ((piece ^ (r & 7)) << 2));
                            const float4 f1 = *reinterpret_cast<const float4 *>(b + r * BK + (((piece + 1) ^ (r & 7)) << 2));
                            split3x8(f0, f1, wh[ni], wm[ni], wl[ni]);
                        }
                        issue_part(ic, 2 * kb2);
                        issue_part(ic, 2 * kb2 + 1);
                        // six partial products, smallest first; W piece first (swapped operands, float4 epilogue)
#define GNNB_DMA_BF6(WP, AP)                                                                                       \
    _Pragma("unroll") for (int mi = 0; mi < MC; mi++) _Pragma("unroll") for (int ni = 0; ni < NT; ni++)             \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(WP[ni]), as_bf16x8(AP[mi]), acc[mi][ni], 0, 0, 0);
                        GNNB_DMA_BF6(wm, am)
                        GNNB_DMA_BF6(wh, al)
                        GNNB_DMA_BF6(wl, ah)
                        GNNB_DMA_BF6(wh, am)
                        GNNB_DMA_BF6(wm, ah)
                        GNNB_DMA_BF6(wh, ah)
#undef GNNB_DMA_BF6
                    }
                } else {
                // (requesting the fragments of k step j + 1 before the MFMAs of step j -- two register sets -- was
                // measured: 601 vs 583 us at the C4 shape)
#pragma unroll
                for (int kb = 0; kb < BK; kb += 8) {
                    float4 fa[MC > 0 ? MC : 1], fb[NT];
                    const int piece = (kb >> 2) + lh; // 16-B piece holding k = kb + 4 lh .. + 3
#pragma unroll
                    for (int mi = 0; mi < MC; mi++) {
                        const int r = rbase + mi * 32 + li;
                        fa[mi] = *reinterpret_cast<const float4 *>(a + r * BK + ((piece ^ (r & 7)) << 2));
                    }
                    if (scaled) { // the row scaler multiplies the A operand, as in the register-staged kernel
#pragma unroll
                        for (int mi = 0; mi < MC; mi++)
                            fa[mi].x *= s[mi], fa[mi].y *= s[mi], fa[mi].z *= s[mi], fa[mi].w *= s[mi];
                    }
#pragma unroll
                    for (int ni = 0; ni < NT; ni++) {
                        const int r = wn * 32 * NT + ni * 32 + li;
                        fb[ni] = *reinterpret_cast<const float4 *>(b + r * BK + ((piece ^ (r & 7)) << 2));
                    }
                    issue_part(ic, kb / 8); // (behind this step's fragment reads, in front of its MFMAs)
                    // operands SWAPPED (W fragment first): the 32x32 accumulator then holds, per lane, FOUR
                    // CONSECUTIVE output columns of one row per register group -- the epilogue stores float4
#pragma unroll
                    for (int mi = 0; mi < MC; mi++)
#pragma unroll
                        for (int ni = 0; ni < NT; ni++) {
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni].x, fa[mi].x, acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni].y, fa[mi].y, acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni].z, fa[mi].z, acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni].w, fa[mi].w, acc[mi][ni], 0, 0, 0);
                        }
                }
                }
            }
            // the marks move on: mk0 = the chunk multiplied next
            if (DNBUF > 2) {
                mk0 = mk1;
                mk1 = issue_end(ic);
            } else {
                mk0 = issue_end(ic);
            }
        }
        if (MC == 0)
            return;

        // D = W_tile . A_tile^T: lane (li, lh) holds Y[row = m_base + li][col = n_base + 8 (reg >> 2) + 4 lh + (reg & 3)]
        auto epilogue = [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int mi = 0; mi < MC; mi++) {
                const int rowg = m0 + rbase + mi * 32 + li;
                if (rowg >= M)
                    continue;
#pragma unroll
                for (int ni = 0; ni < NT; ni++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int colg = n0 + wn * 32 * NT + ni * 32 + 8 * q + 4 * lh;
                        if (vec && colg + 3 < N) {
                            float4 v = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2],
                                                   acc[mi][ni][4 * q + 3]);
                            if (bias) {
                                const float4 bv = *reinterpret_cast<const float4 *>(bias + colg);
                                v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                            }
                            if (skip) {
                                const float4 sk = *reinterpret_cast<const float4 *>(skip + (size_t)rowg * N + colg);
                                v.x += sk.x, v.y += sk.y, v.z += sk.z, v.w += sk.w;
                            }
                            v.x = act_t<ACT>(v.x), v.y = act_t<ACT>(v.y), v.z = act_t<ACT>(v.z), v.w = act_t<ACT>(v.w);
                            *reinterpret_cast<float4 *>(Y + (size_t)rowg * N + colg) = v;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                if (colg + r < N) {
                                    float v = acc[mi][ni][4 * q + r] + (bias ? bias[colg + r] : 0.0f);
                                    if (skip)
                                        v += skip[(size_t)rowg * N + colg + r];
                                    Y[(size_t)rowg * N + colg + r] = act_t<ACT>(v);
                                }
                        }
                    }
            }
        };
        GNNB_DISPATCH_ACT(act, epilogue)
        // the stores just issued sit between the prefetched chunk and the next waits: count them, or the first wait
        // of the next item would drain them.  Only blocks that certainly issued all eight 16-B stores are counted (an
        // under-count merely makes the next waits stricter; an over-count would let a wait return early).
        if (vec && n0 + wn * 64 + 64 <= N) {
#pragma unroll
            for (int mi = 0; mi < MC; mi++)
                if (m0 + rbase + mi * 32 + 32 <= M)
                    vm += 8;
        }
    };

    for (int item = blockIdx.x; item < num_items; item += gridDim.x) {
        int m0, n0, mrows;
        decode(item, m0, n0, mrows);
        const int rpw = max(mrows / (DM / 64), 32); // rows per wave: 64, or 32 in a slice
        const int rbase = wm * rpw;                 // the wave's first row inside the item
        if (rbase >= mrows)                         // (a 32-row slice keeps half of the waves busy)
            run_item(IntTag<0>{}, m0, n0, rbase);
        else if (rpw == 64)
            run_item(IntTag<2>{}, m0, n0, rbase);
        else
            run_item(IntTag<1>{}, m0, n0, rbase);
    }
}

// -------------------------------------------------------------------------------------
// Register-resident-weight variant for K <= 128 (every full-width layer of the d<=128 models, the
// first layer, the MLP head's 64-wide linears).  The weight matrix is tiny next to the activation
// stream, so each wave keeps ITS 32 output columns x K of W in VGPRs for the whole kernel (K/2
// registers) and the workgroup is persistent: it walks a contiguous range of 16-row units, the A
// rows arriving through a double-buffered LDS stage filled by LDS-DMA (global_load_lds) while the
// previous stage is on the matrix cores.  v_mfma_f32_16x16x4_f32 (exact fp32) gives a 16-row
// scheduling quantum, which keeps the persistent ranges balanced.  LDS rows are XOR-swizzled by
// pre-swizzling the DMA *source* address (the DMA destination is lane-linear), which makes the
// ds_read_b128 fragment reads conflict-free:  slot = chunk ^ (row & (P-1)).
// Lane (i = l&15, g = l>>4) reads chunk 4q+g of row i: k = 16q+4g .. +3; MFMA step (q,s) contracts
// k in {16q + 4g + s : g = 0..3}, the same k-permutation on A and W.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef GNNB_LR_SR
#define GNNB_LR_SR 2
#endif
// a full stage's vector epilogue issues 2*SR 16-B stores per wave
#define GNNB_STR2(x) #x
#define GNNB_STR(x) GNNB_STR2(x)
#if GNNB_LR_SR == 1
#define GNNB_LR_NSTORES 2
#elif GNNB_LR_SR == 2
#define GNNB_LR_NSTORES 4
#elif GNNB_LR_SR == 3
#define GNNB_LR_NSTORES 6
#else
#define GNNB_LR_NSTORES 8
#endif
#define GNNB_LR_COUNTED_WAIT "s_waitcnt vmcnt(" GNNB_STR(GNNB_LR_NSTORES) ") lgkmcnt(0)\n\ts_barrier"


// Optional fused gather: when `rec` is set the A stage is not copied from memory but PRODUCED -- the
// workgroup aggregates its destination rows (GCN / sum / mean semantics of k_aggregate_*) from the
// raw feature matrix straight into the LDS stage.  Used for narrow first layers (F_in = 9, 11):
// the gather touches 44-byte rows that live in L2, so the separate aggregate launch and its
// [N, F_in] round trip through memory disappear (reference gcn_conv / gin_conv do the same per
// node: aggregate, then `linear`, gnn_builder_lib.h:1346-1379, :1497-1544).
struct GatherDesc {
    const int4 *rec;     // node records {rp0, deg, j0, j1}{j2, j3, -, -}; nullptr = plain A copy
    const int32_t *col;  // CSR sources (degree > 4)
    const float *dinv;   // GCN normaliser
    int32_t mode;        // gnnb_agg (GCN, SUM, MEAN)
    float eps;
    int32_t cat;         // > 0: the stage row is [aggregate(x)(cat wide) | x_i (cat wide)]  (GraphSAGE: [mean | x], K = 2 cat)
};

// MATH 1 (opt-in, K % 32 == 0, N % 32 == 0, plain A copy): the products go through the bf16 matrix cores as six
// partial products of an exact 3-way split (see split3); A fragments are split in the wave after the LDS read.
template <int KQ, bool VEC_A, int MATH = 0> // KQ = ceil(K/16) in {1,2,4,8}; VEC_A: K % 4 == 0 and 16-B aligned rows
__global__ __launch_bounds__(WG, MATH ? 2 : 3) void k_linear_reg(
    const float *__restrict__ A, int lda, int K, const float *__restrict__ W, int ldw,
    const float *__restrict__ bias, const float *__restrict__ skip, float *__restrict__ Y, int M, int N,
    int act, int rg_log2, int P, int vec_out, GatherDesc gd)
{
    constexpr int SR = GNNB_LR_SR; // 16-row units per stage
    constexpr int EPI_LD = 36; // padded row of the epilogue transpose scratch
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int RG = 1 << rg_log2;     // row groups: waves that take different rows
    const int cw = wave >> rg_log2;  // which 32-column slice this wave owns
    const int rgi = wave & (RG - 1); // which row group
    const int n0 = blockIdx.y * (128 >> rg_log2) + cw * 32;
    const int unit_rows = 16 * RG;
    const int stage_rows = SR * unit_rows;
    const size_t buf_bytes = (((size_t)stage_rows * K * 4) + 15) & ~(size_t)15;
    float *sC = reinterpret_cast<float *>(smem + 2 * buf_bytes) + (size_t)wave * 16 * SR * EPI_LD;

    // ---- persistent range, balanced in UNITS of 16*RG rows (half a stage), so the remainder a
    // workgroup may carry is half a stage.  Local stage j covers units [u0+2j, min(u0+2j+2, u1)).
    const int num_units = (M + unit_rows - 1) / unit_rows;
    const int u0 = (int)(((long long)blockIdx.x * num_units) / gridDim.x);
    const int u1 = (int)(((long long)(blockIdx.x + 1) * num_units) / gridDim.x);
    if (u1 <= u0)
        return;
    const int nstages = (u1 - u0 + SR - 1) / SR;
    const int C = K >> 2; // 16-B chunks per row (VEC_A)
    auto row_begin = [&](int j) { return (u0 + SR * j) * unit_rows; };
    auto rows_of = [&](int j) { return min(min(u0 + SR * j + SR, u1) * unit_rows, M) - (u0 + SR * j) * unit_rows; };

    // ---- this wave's weight slice -> registers
    float breg[2][KQ * 4];
    constexpr int KB = KQ / 2 > 0 ? KQ / 2 : 1; // 32-wide k blocks (MATH 1)
    u32x4 wh[2][KB], wm[2][KB], wl_[2][KB];
    // fast path (wave-uniform): the 32 x K slice is in range and 16-B aligned.  Its rows are read
    // whole (coalesced LDS-DMA) into this wave's share of the not-yet-used stage buffers and picked
    // apart into fragments from LDS; fragment-shaped global loads (16 rows x 64 B per instruction)
    // took ~2 us per workgroup and serialised co-resident workgroups' start.
    const bool wfast = VEC_A && (ldw % 4 == 0) && (K == 16 * KQ) && (n0 + 32 <= N) && (((uintptr_t)W & 15) == 0);
    if (wfast) {
        float *wl = reinterpret_cast<float *>(smem) + (size_t)wave * 16 * K; // 4 x 16*K floats <= 2 buffers
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int nrow0 = n0 + 16 * u;
            const int nch = 16 * C;
            for (int c0 = 0; c0 < nch; c0 += 64) {
                const int L = c0 + lane;
                if (L < nch) {
                    const int rr = L / C, cc = L - rr * C;
                    dma16_to_lds(W + (size_t)(nrow0 + rr) * ldw + cc * 4, reinterpret_cast<char *>(wl) + (size_t)c0 * 16);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // own DMA, wave-private region: no barrier
            if (MATH) { // lane (li, lg) of a 16x16x32 MFMA holds k = 32 kb + 8 lg .. + 7 of column li
#pragma unroll
                for (int kb = 0; kb < KB; kb++) {
                    const float4 f0 = *reinterpret_cast<const float4 *>(wl + (size_t)li * K + 32 * kb + 8 * lg);
                    const float4 f1 = *reinterpret_cast<const float4 *>(wl + (size_t)li * K + 32 * kb + 8 * lg + 4);
                    split3x8(f0, f1, wh[u][kb], wm[u][kb], wl_[u][kb]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < KQ; q++) {
                    const float4 v = *reinterpret_cast<const float4 *>(wl + (size_t)li * K + 16 * q + 4 * lg);
                    breg[u][q * 4 + 0] = v.x;
                    breg[u][q * 4 + 1] = v.y;
                    breg[u][q * 4 + 2] = v.z;
                    breg[u][q * 4 + 3] = v.w;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // fragments read before the region is reused
        }
    } else {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int n = n0 + 16 * u + li;
#pragma unroll
            for (int q = 0; q < KQ; q++) {
                const int k = 16 * q + 4 * lg;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < N)
                    v = load4_guard(W + (size_t)n * ldw + k, K - k, VEC_A && (ldw % 4 == 0));
                breg[u][q * 4 + 0] = v.x;
                breg[u][q * 4 + 1] = v.y;
                breg[u][q * 4 + 2] = v.z;
                breg[u][q * 4 + 3] = v.w;
            }
        }
    }
    float bv[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int n = n0 + 16 * u + li;
        bv[u] = (bias != nullptr && n < N) ? bias[n] : 0.0f;
    }
    // loop-invariant epilogue operands, loaded ONCE: a global load inside the stage loop would make
    // its s_waitcnt also wait for the next stage's DMA (VM operations retire in order)
    const int c4 = (lane & 7) * 4;
    const int nq = n0 + c4;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_out && bias != nullptr && nq < N)
        bq = *reinterpret_cast<const float4 *>(bias + nq);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq.x), "+v"(bq.y), "+v"(bq.z), "+v"(bq.w), "+v"(bv[0]), "+v"(bv[1])::"memory");
    __syncthreads(); // every wave is out of the stage buffers (weight prologue) before A lands there

    auto issue = [&](int j, int bb) {
        char *dst = smem + (size_t)bb * buf_bytes;
        const int m0i = row_begin(j);
        const int rows = rows_of(j);
        if (VEC_A) {
            const int nchunks = rows * C;
            for (int c0 = wave * 64; c0 < nchunks; c0 += 4 * 64) {
                const int L = c0 + lane;
                if (L < nchunks) {
                    const int i = L / C, sl = L - i * C;
                    const int c = sl ^ (i & (P - 1));
                    dma16_to_lds_u(A + (size_t)(m0i + i) * lda + c * 4, dst + (size_t)c0 * 16);
                }
            }
        } else {
            const int nd = rows * K;
            for (int c0 = wave * 64; c0 < nd; c0 += 4 * 64) {
                const int L = c0 + lane;
                if (L < nd) {
                    const int i = L / K, kk = L - i * K;
                    dma4_to_lds_u(A + (size_t)(m0i + i) * lda + kk, dst + (size_t)c0 * 4);
                }
            }
        }
    };

#ifdef GNNB_PROBE
    unsigned long long pt_wait = 0, pt_mma = 0, pt_epi = 0, pt0 = clock64(), pw0 = wall_clock64();
#define GNNB_PT(var, since) do { const unsigned long long _n = clock64(); var += _n - since; since = _n; } while (0)
    unsigned long long pt_last = pt0;
#else
#define GNNB_PT(var, since) do { } while (0)
#endif
    // Stores count in vmcnt on CDNA4 and VM operations retire in order.  A full stage's vector
    // epilogue issues EXACTLY four 16-B stores per wave after the next stage's DMA, so waiting for
    // vmcnt <= 4 proves that DMA has landed while the stores stay in flight; anything irregular
    // (ragged stage, scalar epilogue, a wave without columns) falls back to a full drain.
    const bool wave_has_cols = nq < N || (n0 < N); // some lane of this wave stores
    const bool gather = !VEC_A && gd.rec != nullptr; // workgroup-uniform
    // gather producer: element (row i, feature f) of stage j, neighbours in CSR order, self term last
    auto produce = [&](int j, int bb) {
        float *dst = reinterpret_cast<float *>(smem + (size_t)bb * buf_bytes);
        const int m0i = row_begin(j);
        const int rows = rows_of(j);
        for (int e = tid; e < rows * K; e += WG) {
            const int i = e / K, fk = e - i * K;
            const int node = m0i + i;
            // (GraphSAGE form: columns [0, cat) hold the aggregate, columns [cat, 2 cat) the node's own row)
            const bool own = gd.cat > 0 && fk >= gd.cat;
            const int f = own ? fk - gd.cat : fk;
            const int4 r0 = gd.rec[2 * (size_t)node], r1 = gd.rec[2 * (size_t)node + 1];
            const int deg = r0.y;
            const int jn[4] = {r0.z, r0.w, r1.x, r1.y};
            const float xs = A[(size_t)node * lda + f];
            float xv[4], sv[4];
            const float di = gd.mode == GNNB_AGG_GCN ? gd.dinv[node] : 1.0f;
#pragma unroll
            for (int q = 0; q < 4; q++) { // unused slots alias the node itself (cache hit, discarded)
                xv[q] = A[(size_t)jn[q] * lda + f];
                sv[q] = gd.mode == GNNB_AGG_GCN ? gd.dinv[jn[q]] : 1.0f;
            }
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (deg > q)
                    acc += xv[q] * (di * sv[q]);
            for (int k = r0.x + 4; k < r0.x + deg; k++) {
                const int jj = gd.col[k];
                acc += A[(size_t)jj * lda + f] * (di * (gd.mode == GNNB_AGG_GCN ? gd.dinv[jj] : 1.0f));
            }
            if (gd.mode == GNNB_AGG_GCN)
                acc += xs * (di * di);
            else if (gd.mode == GNNB_AGG_SUM)
                acc += xs * (1.0f + gd.eps);
            else if (deg > 0)
                acc = acc / (float)deg;
            if (own)
                acc = xs;
            dst[e] = acc;
        }
    };
    bool prev_counted = false;
    if (gather)
        produce(0, 0);
    else
        issue(0, 0);
    int b = 0;
    for (int j = 0; j < nstages; j++, b ^= 1) {
        if (prev_counted)
            asm volatile(GNNB_LR_COUNTED_WAIT ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (j + 1 < nstages) {
            if (gather)
                produce(j + 1, b ^ 1); // plain loads + ds_write; the next barrier publishes it
            else
                issue(j + 1, b ^ 1);
        }
        GNNB_PT(pt_wait, pt_last);
        const float *sA = reinterpret_cast<const float *>(smem + (size_t)b * buf_bytes);
        const int m0 = row_begin(j);
        const int m_end = m0 + rows_of(j); // rows past it belong to another workgroup (or nobody)

        f32x4 acc[SR][2];
#pragma unroll
        for (int rt = 0; rt < SR; rt++)
#pragma unroll
            for (int u = 0; u < 2; u++)
                acc[rt][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if (MATH) {
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                u32x4 ah[SR], am[SR], al[SR];
#pragma unroll
                for (int rt = 0; rt < SR; rt++) {
                    const int row = (rt * RG + rgi) * 16 + li;
                    const int c0 = 8 * kb + 2 * lg; // float4 chunks 8 kb + 2 lg, + 1 of the row
                    const float4 f0 = *reinterpret_cast<const float4 *>(sA + (size_t)row * K + ((c0 ^ (row & (P - 1))) << 2));
                    const float4 f1 = *reinterpret_cast<const float4 *>(sA + (size_t)row * K + (((c0 + 1) ^ (row & (P - 1))) << 2));
                    split3x8(f0, f1, ah[rt], am[rt], al[rt]);
                }
                // six partial products, smallest first; the four accumulators interleaved
#define GNNB_BF6(APIECE, BPIECE)                                                                                  \
    _Pragma("unroll") for (int rt = 0; rt < SR; rt++) _Pragma("unroll") for (int u = 0; u < 2; u++)                \
        acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(APIECE[rt]), as_bf16x8(BPIECE[u][kb]), acc[rt][u], 0, 0, 0);
                GNNB_BF6(am, wm)
                GNNB_BF6(al, wh)
                GNNB_BF6(ah, wl_)
                GNNB_BF6(am, wh)
                GNNB_BF6(ah, wm)
                GNNB_BF6(ah, wh)
#undef GNNB_BF6
            }
        } else {
#pragma unroll
            for (int q = 0; q < KQ; q++) {
                float4 a[SR];
    #pragma unroll
                for (int rt = 0; rt < SR; rt++) {
                    const int row = (rt * RG + rgi) * 16 + li; // row inside the stage: unit rt, row group rgi
                    if (VEC_A) {
                        const int c = 4 * q + lg;
                        a[rt] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (c < C)
                            a[rt] = *reinterpret_cast<const float4 *>(sA + (size_t)row * K + ((c ^ (row & (P - 1))) << 2));
                    } else {
                        const int k = 16 * q + 4 * lg;
                        const float *pr = sA + (size_t)row * K + k;
                        a[rt].x = (k + 0 < K) ? pr[0] : 0.f;
                        a[rt].y = (k + 1 < K) ? pr[1] : 0.f;
                        a[rt].z = (k + 2 < K) ? pr[2] : 0.f;
                        a[rt].w = (k + 3 < K) ? pr[3] : 0.f;
                    }
                }
                // k-step outermost: consecutive MFMAs hit the four different accumulators, so the 40-cycle
                // dependent latency of v_mfma_f32_16x16x4_f32 hides behind its 32-cycle issue interval
                float as[SR][4];
    #pragma unroll
                for (int rt = 0; rt < SR; rt++) {
                    as[rt][0] = a[rt].x;
                    as[rt][1] = a[rt].y;
                    as[rt][2] = a[rt].z;
                    as[rt][3] = a[rt].w;
                }
    #pragma unroll
                for (int sk = 0; sk < 4; sk++)
    #pragma unroll
                    for (int rt = 0; rt < SR; rt++)
    #pragma unroll
                        for (int u = 0; u < 2; u++)
                            acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[rt][sk], breg[u][q * 4 + sk], acc[rt][u], 0, 0, 0);
            }
        }
#ifdef GNNB_PROBE
        asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[SR - 1][1][3]));
#endif
        GNNB_PT(pt_mma, pt_last);
        // epilogue: C/D of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
        const bool full = (m_end - m0) == stage_rows;
        prev_counted = vec_out && full && wave_has_cols && (skip == nullptr) && !gather;
        auto epilogue = [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
            if (vec_out) {
                // transpose the wave's 32x32 block through its LDS scratch, then 4 x (ds_read_b128 +
                // 16-B global store) instead of 16 dword stores: 8 lanes cover one 128-B row segment
#pragma unroll
                for (int rt = 0; rt < SR; rt++)
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            sC[(rt * 16 + lg * 4 + r) * EPI_LD + u * 16 + li] = acc[rt][u][r];
                // (same wave wrote and reads: the compiler's lgkmcnt wait orders it; no barrier)
#pragma unroll
                for (int ps = 0; ps < 2 * SR; ps++) {
                    const int rl = ps * 8 + (lane >> 3); // row inside the wave's 16*SR (unit rl>>4)
                    const int m = m0 + ((rl >> 4) * RG + rgi) * 16 + (rl & 15);
                    float4 v = *reinterpret_cast<const float4 *>(sC + rl * EPI_LD + c4);
                    if (m < m_end && nq < N) {
                        v.x += bq.x;
                        v.y += bq.y;
                        v.z += bq.z;
                        v.w += bq.w;
                        if (skip) {
                            const float4 sk = *reinterpret_cast<const float4 *>(skip + (size_t)m * N + nq);
                            v.x += sk.x;
                            v.y += sk.y;
                            v.z += sk.z;
                            v.w += sk.w;
                        }
                        v.x = act_t<ACT>(v.x);
                        v.y = act_t<ACT>(v.y);
                        v.z = act_t<ACT>(v.z);
                        v.w = act_t<ACT>(v.w);
                        *reinterpret_cast<float4 *>(Y + (size_t)m * N + nq) = v;
                    }
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < SR; rt++)
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int n = n0 + 16 * u + li;
                        if (n >= N)
                            continue;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int m = m0 + (rt * RG + rgi) * 16 + lg * 4 + r;
                            if (m < m_end) {
                                float v = acc[rt][u][r] + bv[u];
                                if (skip)
                                    v += skip[(size_t)m * N + n];
                                Y[(size_t)m * N + n] = act_t<ACT>(v);
                            }
                        }
                    }
            }
        };
        GNNB_DISPATCH_ACT(act, epilogue)
        GNNB_PT(pt_epi, pt_last);
    }
#ifdef GNNB_PROBE
    if (tid == 0 && blockIdx.x < 8192 && blockIdx.y == 0) {
        unsigned long long *o = g_probe + blockIdx.x * 8;
        o[0] = pw0;
        o[1] = wall_clock64();
        o[2] = pt_wait;
        o[3] = pt_mma;
        o[4] = pt_epi;
        o[5] = clock64() - pt0;
        o[6] = (unsigned long long)nstages;
    }
#endif
}

template <int KQ, bool VEC_A, int MATH = 0>
static hipError_t launch_linear_reg_t(const float *A, int lda, int K, const float *W, int ldw,
                                      const float *bias, const float *skip, float *Y, int M, int N,
                                      int act, hipStream_t s, const GatherDesc &gd = GatherDesc{})
{
    if (MATH == 0 && VEC_A && KQ >= 2 && options().math == 1 && K == 16 * KQ && N % 32 == 0 && ldw % 4 == 0 &&
        (((uintptr_t)W & 15) == 0) && gd.rec == nullptr)
        return launch_linear_reg_t<KQ, VEC_A, 1>(A, lda, K, W, ldw, bias, skip, Y, M, N, act, s, gd);
    // waves: N <= 32 -> 4 row groups x 1 column slice; N <= 64 -> 2 x 2; else 1 x 4 (128 cols / WG)
    const int rg_log2 = N <= 32 ? 2 : (N <= 64 ? 1 : 0);
    const int cols_per_wg = 128 >> rg_log2;
    const int stage_rows = (16 * GNNB_LR_SR) << rg_log2;
    const int gy = (N + cols_per_wg - 1) / cols_per_wg;
    const size_t buf = (((size_t)stage_rows * K * 4) + 15) & ~(size_t)15;
    const size_t lds = 2 * buf + 4 * 16 * GNNB_LR_SR * 36 * 4; // two stage buffers + per-wave epilogue scratch
    const int vec_out = (N % 4 == 0) && (((uintptr_t)Y & 15) == 0) && (bias == nullptr || ((uintptr_t)bias & 15) == 0) &&
                        (skip == nullptr || ((uintptr_t)skip & 15) == 0);
    int P = 1;
    if (VEC_A) {
        const int C = K / 4;
        while (P < 16 && C % (2 * P) == 0)
            P *= 2;
    }
    const int num_stages = (M + stage_rows - 1) / stage_rows;
    auto kern = k_linear_reg<KQ, VEC_A, MATH>;
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
        if (e != hipSuccess)
            return e;
    }
    // persistent grid = what is resident at once (registers + LDS), asked of the runtime once per
    // LDS size and capped (MI355X_MICROARCH: keep <= 4 blocks of 256 threads per CU)
    static size_t occ_lds = (size_t)-1;
    static int occ_blocks = 1, num_cus = 256;
    if (occ_lds != lds) {
        int nb = 0, devid = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, WG, lds) != hipSuccess || nb < 1)
            nb = 1;
        if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess)
            num_cus = prop.multiProcessorCount;
        occ_blocks = nb;
        occ_lds = lds;
    }
    // K = 128 keeps 64 weight registers per lane and is MFMA-bound: 2 workgroups per CU measured
    // best.  Narrow K is store- / gather-latency-bound: the more resident workgroups the better.
    const int cap = KQ >= 8 ? options().gemm_max_wg_per_cu : (KQ >= 4 ? 3 : 6);
    int gx = num_cus * (occ_blocks > cap ? cap : occ_blocks) / gy;
    if (gx < 1)
        gx = 1;
    if (gx > num_stages)
        gx = num_stages; // at least one full stage per workgroup
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(WG), lds, s, A, lda, K, W, ldw, bias, skip, Y, M, N, act,
                       rg_log2, P, vec_out, gd);
    return hipGetLastError();
}

static bool linear_reg_eligible(const GemmArgs &g)
{
    return options().gemm_variant == 0 && g.nseg == 1 && g.rs[0] == nullptr && g.k[0] <= 128;
}

static hipError_t launch_linear_reg(const GemmArgs &g, const float *w, int ldw, const float *bias,
                                    const float *skip, float *y, int M, int N, int act, hipStream_t s)
{
    const int K = g.k[0];
    const bool vec = g.avec[0] != 0;
    const int kq = K <= 16 ? 1 : (K <= 32 ? 2 : (K <= 64 ? 4 : 8));
#define GNNB_LR_CASE(Q)                                                                              \
    case Q:                                                                                          \
        return vec ? launch_linear_reg_t<Q, true>(g.a[0], g.lda[0], K, w, ldw, bias, skip, y, M, N, act, s) \
                   : launch_linear_reg_t<Q, false>(g.a[0], g.lda[0], K, w, ldw, bias, skip, y, M, N, act, s);
    switch (kq) {
        GNNB_LR_CASE(1)
        GNNB_LR_CASE(2)
        GNNB_LR_CASE(4)
        GNNB_LR_CASE(8)
    }
#undef GNNB_LR_CASE
    return hipErrorInvalidValue;
}

// -------------------------------------------------------------------------------------
// k_linear_wlds: the K, N <= 128 dense update with the WEIGHTS IN LDS and no workgroup barrier in the loop.
// Reference: `linear` per node vector (gnn_builder_lib.h:808-905); here Y[M,N] = act(A[M,K] . W[N,K]^T + b (+ skip)).
//
// What the probe of k_linear_reg showed (profiles/r02_linear_reg_probe.txt): of a wave's cycles 60 % are the MFMA
// loop (two waves of a SIMD compete for one pipe), 23 % the per-stage barrier (four waves on four SIMDs, each
// sharing its SIMD with a wave of another workgroup, arrive skewed) and 16 % the epilogue (transpose through LDS).
// Here every WAVE is independent:
//   * W (<= 64 KB) is loaded ONCE per workgroup into LDS (LDS-DMA, XOR-swizzled through the source address) and
//     only read afterwards -- no synchronisation after the prologue;
//   * each wave streams its own 16-row units of A through a private LDS ring (untracked LDS-DMA, counted vmcnt
//     waits as in the gather-aggregate ring), so a slow wave delays nobody;
//   * the MFMA operands are SWAPPED (W fragment as the A operand): the 16x16 accumulator then holds
//     Y[m0 + li][n0 + 4 lg .. + 3] per lane, i.e. four CONSECUTIVE output columns -- bias / skip / activation are
//     float4 operations and the result is stored with one 16-B store per tile, no transpose;
//   * one wave per SIMD (four per CU): the fp32 matrix pipe has a single client that issues back to back, with the
//     next k block's fragments requested from LDS before the current block's 4 NT MFMAs are issued.
// Eligibility: K, N in {64, 128}, 16-B aligned rows; anything else takes k_linear_reg / k_linear.
template <int KQ, int NT, int ACT>
__global__ __launch_bounds__(WG, 1) void k_linear_wlds(const float *__restrict__ A, int lda, const float *__restrict__ W,
                                                      int ldw, const float *__restrict__ bias, float *__restrict__ Y,
                                                      int M, int nslots)
{
    constexpr int K = 16 * KQ, N = 16 * NT;
    constexpr int C = K / 4;                           // 16-B chunks per A / W row
    constexpr int P = C >= 16 ? 16 : C;                // XOR-swizzle period (power of two)
    constexpr int SLOT = 16 * K * 4;                   // one 16-row unit of A; its DMA is exactly KQ wave-instructions
    constexpr int TPB = (NT + KQ - 1) / KQ;            // deferred stores issued per k block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    char *wl = smem;                                   // W: N rows x K floats, swizzled
    char *ring = smem + N * K * 4 + (size_t)wave * nslots * SLOT;

    // ---- this wave's run of 16-row units
    const int num_units = (M + 15) >> 4;
    const int gw = blockIdx.x * (WG / 64) + wave, tw = gridDim.x * (WG / 64);
    const int u0 = (int)(((long long)gw * num_units) / tw), u1 = (int)(((long long)(gw + 1) * num_units) / tw);

    // ---- prologue: the whole W -> LDS, all four waves; chunk sl of row n lands in slot sl, holding source chunk sl ^ (n & (P-1))
    for (int c0 = wave * 64; c0 < N * C; c0 += WG) {
        const int L = c0 + lane;
        if (L < N * C) {
            const int n = L / C, sl = L - n * C;
            dma16_to_lds_u(W + (size_t)n * ldw + ((sl ^ (n & (P - 1))) << 2), wl + (size_t)c0 * 16);
        }
    }
    float4 bq[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
        bq[t] = bias ? *reinterpret_cast<const float4 *>(bias + 16 * t + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < NT; t++)
        asm volatile("" : "+v"(bq[t].x), "+v"(bq[t].y), "+v"(bq[t].z), "+v"(bq[t].w)); // loaded HERE, not inside the loop

    // one 1-KiB piece (64 chunks) of unit u's A rows -> its slot; returns 1 if the instruction was issued
    const int lrow = lane / C, lsl = lane - lrow * C;  // (C >= 16: a piece covers 64 / C whole rows)
    auto issue_piece = [&](int u, int slot, int q) -> int {
        const int m0 = u << 4;
        const int rows = min(16, M - m0);
        if (q * 64 >= rows * C)
            return 0; // wave-uniform
        const int i = q * (64 / C) + lrow;
        if (i < rows)
            dma16_to_lds_u(A + (size_t)(m0 + i) * lda + ((lsl ^ (i & (P - 1))) << 2), ring + (size_t)slot * SLOT + (size_t)q * 1024);
        return 1;
    };

    // ring bookkeeping: unit u lives in slot (u - u0) % nslots; f_mark[j] = VM operations issued when the DMA of the
    // j-th oldest outstanding unit was complete
    int vm = 0;
    int f_mark[4] = {0, 0, 0, 0};
    const int ahead = nslots - 1; // units requested before they are needed
    for (int j = 0; j < (ahead > 0 ? ahead : 1) && u0 + j < u1; j++) {
#pragma unroll
        for (int q = 0; q < KQ; q++)
            vm += issue_piece(u0 + j, j, q);
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i == j)
                f_mark[i] = vm;
    }
    // W (and the first units) landed for every wave: the only barrier of the kernel
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    const float *wls = reinterpret_cast<const float *>(wl);
    const int swz = li & (P - 1); // (16 t + li) & (P - 1) == li & (P - 1): one swizzle term for W rows and A rows
#ifdef GNNB_PROBE
    unsigned long long pt_wait = 0, pt_mma = 0, pt_epi = 0, pt0 = clock64(), pw0 = wall_clock64(), pt_last = pt0;
#endif
    float4 pv[NT];           // the previous unit's finished tiles: stored during THIS unit's MFMA stream
    int pm = M;              // ... their row (>= M: nothing to store)
    bool have_prev = false;  // wave-uniform
    int head_slot = 0, fill_slot = ahead == 0 ? 0 : (ahead % nslots);
    auto store_prev = [&](int t) { // tile t of the previous unit
        if (pm < M)
            *reinterpret_cast<float4 *>(Y + (size_t)pm * N + 16 * t + 4 * lg) = pv[t];
    };
    for (int u = u0; u < u1; u++) {
        vmcnt_wait_n(min(vm - f_mark[0], 63));
        GNNB_PT(pt_wait, pt_last);
        const float *sa = reinterpret_cast<const float *>(ring + (size_t)head_slot * SLOT);
        const int un = u + ahead;                 // the unit requested during this one (into the slot freed last time)
        const bool more = ahead > 0 && un < u1;
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++)
            acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto afrag = [&](int q) { return *reinterpret_cast<const float4 *>(sa + li * K + (((4 * q + lg) ^ swz) << 2)); };
        auto wfrag = [&](int q, int t) {
            return *reinterpret_cast<const float4 *>(wls + (16 * t + li) * K + (((4 * q + lg) ^ swz) << 2));
        };
        // Two fragment sets, statically alternated (the q loop is fully unrolled).  The scheduler barriers pin the
        // order "request block q+1's nine fragments, THEN issue block q's 4 NT MFMAs": left alone the compiler sinks
        // every ds_read to just above its first use and the single wave of the SIMD eats the LDS latency nine
        // times per k block (measured: 2x).  The previous unit's stores and the next unit's DMA pieces ride in the
        // same stream, one piece per k block: a vector-memory issue costs the wave 60-180 cycles, which the matrix
        // pipe spends on the MFMAs already queued.
        float4 af[2], wf[2][NT];
        af[0] = afrag(0);
#pragma unroll
        for (int t = 0; t < NT; t++)
            wf[0][t] = wfrag(0, t);
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            const int cb = q & 1, nb2 = cb ^ 1;
            if (q + 1 < KQ) {
                af[nb2] = afrag(q + 1);
#pragma unroll
                for (int t = 0; t < NT; t++)
                    wf[nb2][t] = wfrag(q + 1, t);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (have_prev) {
#pragma unroll
                for (int i = 0; i < TPB; i++)
                    if (q * TPB + i < NT) {
                        store_prev(q * TPB + i);
                        vm++;
                    }
            }
            if (more)
                vm += issue_piece(un, fill_slot, q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sk = 0; sk < 4; sk++) {
                const float av = sk == 0 ? af[cb].x : (sk == 1 ? af[cb].y : (sk == 2 ? af[cb].z : af[cb].w));
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const float wv = sk == 0 ? wf[cb][t].x : (sk == 1 ? wf[cb][t].y : (sk == 2 ? wf[cb][t].z : wf[cb][t].w));
                    // operands swapped: D[n][m] -- the lane ends up with Y[m0 + li][16 t + 4 lg + r], r = 0..3
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, av, acc[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef GNNB_PROBE
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[NT - 1][3]));
#endif
        GNNB_PT(pt_mma, pt_last);
        // ---- epilogue (VALU only): bias + activation on float4; the stores follow inside the next unit's stream
#pragma unroll
        for (int t = 0; t < NT; t++) {
            pv[t].x = act_t<ACT>(acc[t][0] + bq[t].x);
            pv[t].y = act_t<ACT>(acc[t][1] + bq[t].y);
            pv[t].z = act_t<ACT>(acc[t][2] + bq[t].z);
            pv[t].w = act_t<ACT>(acc[t][3] + bq[t].w);
        }
        pm = (u << 4) + li;
        have_prev = true;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this slot's LDS reads are done before it is refilled
        // retire unit u; the unit requested during it joins the tail of the queue
#pragma unroll
        for (int i = 0; i + 1 < 4; i++)
            f_mark[i] = f_mark[i + 1];
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i == ahead - 1)
                    f_mark[i] = vm;
        }
        if (ahead == 0 && u + 1 < u1) { // single slot: no overlap, request the next unit now
#pragma unroll
            for (int q = 0; q < KQ; q++)
                vm += issue_piece(u + 1, 0, q);
            f_mark[0] = vm;
        }
        fill_slot = head_slot; // the slot just consumed is the next to be refilled
        head_slot = head_slot + 1 == nslots ? 0 : head_slot + 1;
        GNNB_PT(pt_epi, pt_last);
    }
    if (have_prev) {
#pragma unroll
        for (int t = 0; t < NT; t++)
            store_prev(t);
    }
#ifdef GNNB_PROBE
    if (lane == 0 && wave == 0 && blockIdx.x < 8192) {
        unsigned long long *o = g_probe + blockIdx.x * 8;
        o[0] = pw0;
        o[1] = wall_clock64();
        o[2] = pt_wait;
        o[3] = pt_mma;
        o[4] = pt_epi;
        o[5] = clock64() - pt0;
        o[6] = (unsigned long long)(u1 - u0);
    }
#endif
}

static bool linear_wlds_eligible(const GemmArgs &g, const float *w, int ldw, const float *bias, const float *skip,
                                 const float *y, int N)
{
    if (options().gemm_variant != 0 || options().math != 0 || !options().gemm_wlds)
        return false;
    if (skip != nullptr) // (a skip tile per unit would not leave room for the ring beside a 64 KB W: k_linear_reg)
        return false;
    if (g.nseg != 1 || g.rs[0] != nullptr || !g.avec[0])
        return false;
    const int K = g.k[0];
    if (!(K == 64 || K == 128) || !(N == 64 || N == 128))
        return false;
    return (ldw % 4 == 0) && (((uintptr_t)w & 15) == 0) && (((uintptr_t)y & 15) == 0) &&
           (bias == nullptr || ((uintptr_t)bias & 15) == 0) && (skip == nullptr || ((uintptr_t)skip & 15) == 0);
}

static hipError_t launch_linear_wlds(const GemmArgs &g, const float *w, int ldw, const float *bias, const float *skip,
                                     float *y, int M, int N, int act, hipStream_t s)
{
    const int K = g.k[0];
    static int num_cus = 0;
    if (num_cus == 0) {
        int devid = 0;
        hipDeviceProp_t prop;
        num_cus = (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess)
                      ? prop.multiProcessorCount : 256;
    }
    const int slot = 16 * K * 4;
    // ring depth: what fits beside W in the CU's 160 KiB of LDS, at most 4.  A unit is requested ns - 1 units before
    // it is needed, piece by piece inside the MFMA stream.  Measured (tools/bench_gemm.py): ns = 2 beats ns = 3 at the
    // BASELINE sizes (31.9 vs 35.8 us at M = 73 763): a wave has only 4-7 units, and the deeper ring's longer blocking
    // prologue costs more than its steadier stream gains.
    int ns = (int)((160 * 1024 - (size_t)N * K * 4) / ((size_t)4 * slot));
    ns = std::min(std::max(ns, 1), std::min(options().gemm_wlds_slots, 4));
    const size_t lds = (size_t)N * K * 4 + (size_t)4 * ns * slot;
    const int num_units = (M + 15) / 16;
    int grid = std::min(num_cus, (num_units + 3) / 4);
    if (grid < 1)
        grid = 1;
    hipError_t rc = hipSuccess;
    auto go = [&](auto atag, auto qtag, auto ntag) {
        constexpr int ACT = decltype(atag)::value, KQ = decltype(qtag)::value, NTL = decltype(ntag)::value;
        auto kern = k_linear_wlds<KQ, NTL, ACT>;
        rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
        if (rc != hipSuccess)
            return;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WG), lds, s, g.a[0], g.lda[0], w, ldw, bias, y, M, ns);
        rc = hipGetLastError();
    };
    auto go_k = [&](auto atag) {
        if (K == 128 && N == 128) go(atag, IntTag<8>{}, IntTag<8>{});
        else if (K == 128) go(atag, IntTag<8>{}, IntTag<4>{});
        else if (N == 128) go(atag, IntTag<4>{}, IntTag<8>{});
        else go(atag, IntTag<4>{}, IntTag<4>{});
    };
    GNNB_DISPATCH_ACT(act, go_k)
    return rc;
}

// Fused narrow-input conv: Y = act(aggregate(x) . W^T + b (+ skip)) in one launch (K <= 32).
hipError_t launch_conv_gather(const BatchTables &t, int agg_kind, float eps, const float *x, int lda,
                              int K, const float *w, int ldw, const float *bias, const float *skip,
                              float *y, int N, int act, hipStream_t s, int cat)
{
    // K = width of the stage row the GEMM contracts over: F_in, or 2 F_in in the [aggregate | own row] form
    if (K > 32 || agg_kind == GNNB_AGG_PNA || t.num_nodes <= 0 || (cat > 0 && K != 2 * cat))
        return hipErrorNotSupported;
    GatherDesc gd;
    gd.rec = t.node_rec;
    gd.col = t.col;
    gd.dinv = t.dinv;
    gd.mode = agg_kind;
    gd.eps = eps;
    gd.cat = cat;
    if (K <= 16)
        return launch_linear_reg_t<1, false>(x, lda, K, w, ldw, bias, skip, y, t.num_nodes, N, act, s, gd);
    return launch_linear_reg_t<2, false>(x, lda, K, w, ldw, bias, skip, y, t.num_nodes, N, act, s, gd);
}

hipError_t launch_linear(const GemmArgs &g, const float *w, int ldw, const float *bias,
                         const float *skip, float *y, int M, int N, int act, hipStream_t s)
{
    if (M <= 0 || N <= 0)
        return hipSuccess;
    if (linear_wlds_eligible(g, w, ldw, bias, skip, y, N))
        return launch_linear_wlds(g, w, ldw, bias, skip, y, M, N, act, s);
    if (linear_reg_eligible(g))
        return launch_linear_reg(g, w, ldw, bias, skip, y, M, N, act, s);
    const int gm = (M + BM - 1) / BM;
    if (N > 64 && options().gemm_dma) {
        bool plain = (ldw % 4 == 0) && (((uintptr_t)w & 15) == 0);
        for (int sg = 0; sg < g.nseg && plain; sg++)
            plain = g.avec[sg] && g.wvec[sg] && (g.k[sg] % BK == 0) && (g.koff[sg] % 4 == 0);
        if (plain) {
            const size_t lds = (size_t)DNBUF * DBUF_B;
            {
                hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(options().math ? k_linear_dma<1> : k_linear_dma<0>), lds);
                if (e != hipSuccess)
                    return e;
            }
            static int num_cus = 0;
            if (num_cus == 0) {
                int devid = 0;
                hipDeviceProp_t prop;
                num_cus = (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess)
                              ? prop.multiProcessorCount : 256;
            }
            const int tm = (M + DM - 1) / DM, tn = (N + DN - 1) / DN, tiles = tm * tn;
            // two 64-KB workgroups are resident per CU and share its matrix pipe: what has to come out even is the
            // work per CU.  The last, partial round of tiles (all of them when there are fewer tiles than CUs) goes out
            // in 2 or 4 row slices per tile when those still fit one round (see the kernel)
            const int rem = tiles % num_cus;
            int split = 1;
            if (options().gemm_tail_split && rem > 0)
                split = 4 * rem <= DWGPC * num_cus ? 4 : (2 * rem <= DWGPC * num_cus ? 2 : 1);
            const int split_from = split > 1 ? tiles - rem : tiles;
            const int grid = std::min(split_from + split * (tiles - split_from), DWGPC * num_cus);
            if (options().math)
                hipLaunchKernelGGL(k_linear_dma<1>, dim3(grid), dim3(DWG), lds, s, g, w, ldw, bias, skip, y, M, N, act, tm, tn,
                                   split_from, split);
            else
                hipLaunchKernelGGL(k_linear_dma<0>, dim3(grid), dim3(DWG), lds, s, g, w, ldw, bias, skip, y, M, N, act, tm, tn,
                                   split_from, split);
            return hipGetLastError();
        }
    }
    if (N > 64) {
        constexpr int NT = 2;
        const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * 64 * NT * LDS_LD) * 4;
        {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_linear<NT>), lds);
            if (e != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(k_linear<NT>, dim3(gm, (N + 127) / 128), dim3(WG), lds, s, g, w, ldw,
                           bias, skip, y, M, N, act);
    } else {
        constexpr int NT = 1;
        const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * 64 * NT * LDS_LD) * 4;
        hipLaunchKernelGGL(k_linear<NT>, dim3(gm, 1), dim3(WG), lds, s, g, w, ldw, bias, skip, y, M,
                           N, act);
    }
    return hipGetLastError();
}

// =====================================================================================
// global pooling
// =====================================================================================
// Reference: global_add_pool / global_mean_pool / global_max_pool
// (gnn_builder_lib.h:2709-2739, :2741-2771, :2773-2803) concatenated in `aggrs` order
// (templates/model.cpp.jinja:440-448; gnnbuilder/models.py:348-352).  One lane group
// (d/4 lanes, float4 each) owns one graph and walks its node rows in order, so the sum
// order equals the reference's; all requested reductions come from a single read of x.
template <int VEC>
__global__ __launch_bounds__(WG) void k_global_pool(const float *__restrict__ x,
                                                    const int32_t *__restrict__ node_ptr, int B,
                                                    int d, int glog2, int p0, int p1, int p2,
                                                    int np, float *__restrict__ out)
{
    typedef Vf<VEC> V;
    const int G = 1 << glog2;
    const int grp = threadIdx.x >> glog2;
    const int gl = threadIdx.x & (G - 1);
    const int g = blockIdx.x * (WG >> glog2) + grp;
    if (g >= B)
        return;
    const int n0 = node_ptr[g], n1 = node_ptr[g + 1];
    const int nvec = d / VEC;
    const int pools[3] = {p0, p1, p2};
    for (int f = gl; f < nvec; f += G) {
        const int fo = f * VEC;
        V sum = V::splat(0.0f), mx = V::splat(0.0f);
        int i = n0;
        if (i < n1) {
            const V v = V::load(x + (size_t)i * d + fo);
            sum = v;
            mx = v;
            i++;
        }
        for (; i + 3 < n1; i += 4) {
            const V a = V::load(x + (size_t)i * d + fo);
            const V b = V::load(x + (size_t)(i + 1) * d + fo);
            const V c = V::load(x + (size_t)(i + 2) * d + fo);
            const V e = V::load(x + (size_t)(i + 3) * d + fo);
            sum = vadd(vadd(vadd(vadd(sum, a), b), c), e);
            mx = vmax(vmax(mx, a), vmax(b, vmax(c, e)));
        }
        for (; i < n1; i++) {
            const V v = V::load(x + (size_t)i * d + fo);
            sum = vadd(sum, v);
            mx = vmax(mx, v);
        }
        const int n = n1 - n0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k >= np)
                break;
            V r = sum;
            if (pools[k] == GNNB_POOL_MEAN)
                r = n > 0 ? vdiv(sum, V::splat((float)n)) : V::splat(0.0f);
            else if (pools[k] == GNNB_POOL_MAX)
                r = mx;
            r.store(out + (size_t)g * np * d + (size_t)k * d + fo);
        }
    }
}

hipError_t launch_global_pool(const float *x, const int32_t *node_ptr, int num_graphs, int d,
                              const int32_t *pools, int num_pools, float *out, hipStream_t s)
{
    if (num_graphs <= 0)
        return hipSuccess;
    const bool v4 = (d % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)out & 15) == 0);
    const int nvec = v4 ? d / 4 : d;
    int glog2 = 2;
    while ((1 << glog2) < nvec && glog2 < 6)
        glog2++;
    const int per_wg = WG >> glog2;
    const int grid = (num_graphs + per_wg - 1) / per_wg;
    const int p0 = pools[0], p1 = num_pools > 1 ? pools[1] : 0, p2 = num_pools > 2 ? pools[2] : 0;
    if (v4)
        hipLaunchKernelGGL(k_global_pool<4>, dim3(grid), dim3(WG), 0, s, x, node_ptr, num_graphs, d,
                           glog2, p0, p1, p2, num_pools, out);
    else
        hipLaunchKernelGGL(k_global_pool<1>, dim3(grid), dim3(WG), 0, s, x, node_ptr, num_graphs, d,
                           glog2, p0, p1, p2, num_pools, out);
    return hipGetLastError();
}


// =====================================================================================
// fused readout: global pooling + MLP head
// =====================================================================================
// Reference: compute_global_graph_pooling + compute_mlp_head (templates/model.cpp.jinja:413-530;
// global_*_pool gnn_builder_lib.h:2709-2803; MLP gnnbuilder/models.py:398-430).  As separate
// launches the head is three GEMMs with M = B rows (32 workgroups on a 256-CU chip) behind a
// pooling pass that writes and re-reads [B, k*d].  Here one workgroup owns 16 graphs:
//   0. fires the LDS-DMA of ALL head weights (they fit LDS: 119 KB at the BASELINE configs),
//   1. pools its graphs' node rows (the only HBM-sized read) into a [16, k*d] LDS tile while the
//      weights land -- one lane group per graph, rows in order, so the sum order is the reference's,
//   2. runs every linear as 16 x n MFMA tiles (v_mfma_f32_16x16x4_f32, A and W fragments both
//      from LDS, XOR-swizzled rows), activations staying in LDS,
//   3. writes [16, OUT].
// The pooled tile and the hidden activations never touch HBM.
static constexpr int HEAD_GRAPHS = 16;
static constexpr int HEAD_THREADS = 512; // 8 waves: 16 graphs pooled in parallel (32 lanes each at d=128)

__device__ inline int head_swz_p(int k)
{
    // largest power of two <= 16 dividing the number of 16-B chunks per row (1 = no swizzle)
    if (k & 3)
        return 1;
    const int c = k >> 2;
    int p = 1;
    while (p < 16 && (c % (2 * p)) == 0)
        p *= 2;
    return p;
}
// float offset of element (row, k) in a [rows][kdim] LDS image with 16-B chunks XOR-swizzled
__device__ inline int head_off(int row, int k, int kdim, int P)
{
    return row * kdim + ((((k >> 2) ^ (row & (P - 1))) << 2) | (k & 3));
}

template <int ACT>
__global__ __launch_bounds__(HEAD_THREADS) void k_pool_mlp(const float *__restrict__ x,
                                                 const int32_t *__restrict__ node_ptr, int B, int d,
                                                 int glog2, int p0, int p1, int p2, int np,
                                                 HeadArgs head, float *__restrict__ out,
                                                 const float *__restrict__ prepooled,
                                                 int act0_floats, int act1_floats, int woff0, int woff1, int woff2, int woff3,
                                                 int woff4, int woff5, int woff6, int woff7)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // buffer 0 holds the pooled tile and every even layer's output, buffer 1 the odd ones
    // (LDS pointers derived arithmetically from smem: a runtime-indexed pointer array turns the
    // accesses into FLAT loads that wait on vmcnt)
    float *const act_lo = reinterpret_cast<float *>(smem);
    float *wbase = reinterpret_cast<float *>(smem) + (size_t)act0_floats + act1_floats;
    const int woff[8] = {woff0, woff1, woff2, woff3, woff4, woff5, woff6, woff7};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int g0 = blockIdx.x * HEAD_GRAPHS;

    GNNB_STAMP(0);
    // ---- 0. all weights (+ biases) -> LDS: row-wise linear DMA, swizzle applied on the source side
#pragma unroll 1
    for (int l = 0; l < head.nlin; l++) {
        const int k = head.dims[l], n = head.dims[l + 1];
        const float *W = head.w[l];
        float *dstf = wbase + woff[l];
        if ((k & 3) == 0) {
            const int C = k >> 2, P = head_swz_p(k), nch = n * C;
            for (int c0 = wave * 64; c0 < nch; c0 += (HEAD_THREADS / 64) * 64) {
                const int L = c0 + lane;
                if (L < nch) {
                    const int r = L / C, sl = L - r * C;
                    dma16_to_lds(W + (size_t)r * k + ((sl ^ (r & (P - 1))) << 2),
                                 reinterpret_cast<char *>(dstf) + (size_t)c0 * 16);
                }
            }
        } else {
            const int nd = n * k;
            for (int c0 = wave * 64; c0 < nd; c0 += (HEAD_THREADS / 64) * 64)
                if (c0 + lane < nd)
                    dma4_to_lds(W + c0 + lane, reinterpret_cast<char *>(dstf) + (size_t)c0 * 4);
        }
        // bias right behind its matrix (zeros when the layer has none)
        float *dstb = dstf + (((size_t)n * k + 3) & ~(size_t)3);
        if (head.b[l] != nullptr) {
            for (int c0 = wave * 64; c0 < n; c0 += (HEAD_THREADS / 64) * 64)
                if (c0 + lane < n)
                    dma4_to_lds(head.b[l] + c0 + lane, reinterpret_cast<char *>(dstb) + (size_t)c0 * 4);
        } else {
            for (int i = tid; i < n; i += HEAD_THREADS)
                dstb[i] = 0.0f;
        }
    }
    GNNB_STAMP(1);
    if (prepooled != nullptr) {
        // ---- 1'. the pooled tile already exists ([B, k*d], written by the fused conv stack): DMA this
        // workgroup's 16 rows into the swizzled LDS tile
        const int k0 = head.dims[0], C0 = k0 >> 2, P0 = head_swz_p(k0);
        const int rows = min(HEAD_GRAPHS, B - g0), nch = rows * C0;
        for (int c0 = wave * 64; c0 < nch; c0 += (HEAD_THREADS / 64) * 64) {
            const int L = c0 + lane;
            if (L < nch) {
                const int r = L / C0, sl = L - r * C0;
                dma16_to_lds(prepooled + (size_t)(g0 + r) * k0 + ((sl ^ (r & (P0 - 1))) << 2),
                             reinterpret_cast<char *>(act_lo) + (size_t)c0 * 16);
            }
        }
    } else
    // ---- 1. pooling: one lane group per graph, rows in order
    {
        const int G = 1 << glog2, groups = HEAD_THREADS >> glog2;
        const int grp = tid >> glog2, gl = tid & (G - 1);
        const int k0 = head.dims[0], P0 = head_swz_p(k0);
        const int pools[3] = {p0, p1, p2};
        const int nvec = d >> 2;
        for (int gi = grp; gi < HEAD_GRAPHS; gi += groups) {
            const int g = g0 + gi;
            const int n0 = g < B ? node_ptr[g] : 0, n1 = g < B ? node_ptr[g + 1] : 0;
            for (int f = gl; f < nvec; f += G) {
                const int fo = f * 4;
                typedef Vf<4> V;
                V sum = V::splat(0.0f), mx = V::splat(0.0f);
                int i = n0;
                if (i < n1) {
                    sum = V::load(x + (size_t)i * d + fo);
                    mx = sum;
                    i++;
                }
                for (; i + 7 < n1; i += 8) { // eight independent 16-B loads in flight per lane
                    V r8[8];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        r8[u] = V::load(x + (size_t)(i + u) * d + fo);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        sum = vadd(sum, r8[u]); // row order kept: the reference's sum order
                        mx = vmax(mx, r8[u]);
                    }
                }
                for (; i + 3 < n1; i += 4) {
                    const V a = V::load(x + (size_t)i * d + fo);
                    const V b = V::load(x + (size_t)(i + 1) * d + fo);
                    const V c = V::load(x + (size_t)(i + 2) * d + fo);
                    const V e = V::load(x + (size_t)(i + 3) * d + fo);
                    sum = vadd(vadd(vadd(vadd(sum, a), b), c), e);
                    mx = vmax(vmax(mx, a), vmax(b, vmax(c, e)));
                }
                for (; i < n1; i++) {
                    const V v = V::load(x + (size_t)i * d + fo);
                    sum = vadd(sum, v);
                    mx = vmax(mx, v);
                }
                const int n = n1 - n0;
#pragma unroll
                for (int kk = 0; kk < 3; kk++) {
                    if (kk >= np)
                        break;
                    V r = sum;
                    if (pools[kk] == GNNB_POOL_MEAN)
                        r = n > 0 ? vdiv(sum, V::splat((float)n)) : V::splat(0.0f);
                    else if (pools[kk] == GNNB_POOL_MAX)
                        r = mx;
                    r.store(act_lo + head_off(gi, kk * d + fo, k0, P0));
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F); // (lgkmcnt(0): pooled tile written)
    GNNB_STAMP(2);
    __syncthreads(); // drains vmcnt: weights have landed; pooled tile complete
    GNNB_STAMP(3);

    // ---- 2. the linears
    int cur = 0;
#pragma unroll 1
    for (int l = 0; l < head.nlin; l++) {
        const int k = head.dims[l], n = head.dims[l + 1];
        const bool last = (l == head.nlin - 1);
        const int Pk = head_swz_p(k), Pn = head_swz_p(n);
        const float *sA = reinterpret_cast<const float *>(smem) + (cur ? act0_floats : 0);
        float *sY = reinterpret_cast<float *>(smem) + (cur ? 0 : act0_floats);
        const float *sW = wbase + woff[l];
        const float *sbias = sW + (((size_t)n * k + 3) & ~(size_t)3); // staged next to the matrix
        const bool vec = (k & 3) == 0;
        for (int sl = wave; sl * 16 < n; sl += HEAD_THREADS / 64) {
            const int nn = sl * 16 + li; // this lane's output column (B-fragment row of W)
            // four independent accumulator chains over interleaved 16-wide k blocks: the LDS reads of
            // four blocks are issued together and the MFMAs never wait on each other
            f32x4 accs[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                accs[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            auto frag = [&](int kb, float4 &a, float4 &w) {
                const int kk = kb + 4 * lg;
                a = make_float4(0.f, 0.f, 0.f, 0.f);
                w = a;
                if (vec) {
                    if (kk < k) {
                        a = *reinterpret_cast<const float4 *>(sA + head_off(li, kk, k, Pk));
                        if (nn < n)
                            w = *reinterpret_cast<const float4 *>(sW + head_off(nn, kk, k, Pk));
                    }
                } else {
                    const float *pa = sA + li * k + kk;
                    const float *pw = sW + (size_t)nn * k + kk;
                    a.x = kk + 0 < k ? pa[0] : 0.f;
                    a.y = kk + 1 < k ? pa[1] : 0.f;
                    a.z = kk + 2 < k ? pa[2] : 0.f;
                    a.w = kk + 3 < k ? pa[3] : 0.f;
                    if (nn < n) {
                        w.x = kk + 0 < k ? pw[0] : 0.f;
                        w.y = kk + 1 < k ? pw[1] : 0.f;
                        w.z = kk + 2 < k ? pw[2] : 0.f;
                        w.w = kk + 3 < k ? pw[3] : 0.f;
                    }
                }
            };
            // fast path (wave-uniform): every 64-wide k block is whole -> unguarded loads that are all
            // in flight together (a lane-dependent guard makes the compiler wait at each join); an
            // out-of-range output column re-reads the last valid W row and is dropped at the store
            const bool kfast = vec && (k % 64 == 0);
            const int nnc = nn < n ? nn : n - 1;
            for (int kb = 0; kb < k; kb += 64) {
                float4 a[4], w[4];
                if (kfast) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int kk = kb + 16 * u + 4 * lg;
                        a[u] = *reinterpret_cast<const float4 *>(sA + head_off(li, kk, k, Pk));
                        w[u] = *reinterpret_cast<const float4 *>(sW + head_off(nnc, kk, k, Pk));
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        frag(kb + 16 * u, a[u], w[u]); // blocks past k come back as zeros
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, w[u].x, accs[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, w[u].y, accs[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, w[u].z, accs[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, w[u].w, accs[u], 0, 0, 0);
            }
            f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; r++)
                acc[r] = (accs[0][r] + accs[1][r]) + (accs[2][r] + accs[3][r]);
            // C/D: col = lane&15 (output column nn), row = (lane>>4)*4 + r (graph inside the tile)
            if (nn < n) {
                const float bvv = sbias[nn];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gi = lg * 4 + r;
                    const float v = acc[r] + bvv;
                    if (last) {
                        if (g0 + gi < B)
                            out[(size_t)(g0 + gi) * n + nn] = v;
                    } else {
                        sY[(n & 3) == 0 ? head_off(gi, nn, n, Pn) : gi * n + nn] = act_t<ACT>(v);
                    }
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
#ifdef GNNB_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        g_probe[blockIdx.x * 8 + 1] = wall_clock64(); // overwrite slot-0 cycle word: end time
    }
#endif
}

// -------------------------------------------------------------------------------------
// Readout on a pooled matrix, small-footprint form.  k_pool_mlp keeps every weight in LDS (119 KB at the
// BASELINE configs): fast on an idle chip, but when batches are in flight on several streams it cannot
// start on a CU until BOTH resident workgroups of the next batch's conv-stack kernel have left, and while
// it runs nothing else fits -- measured cost 12.5 us per step at C2 for 10.5 us of kernel.  This form
// needs ~9 KB of LDS and < 96 registers, so its 4-wave workgroups slot in BESIDE the conv-stack kernel
// (31 KB of LDS and one wave slot of 96 registers per SIMD are left over there): weights and the pooled
// rows are MFMA operands fetched straight from L2 (the pooled matrix was just written, the head's weights
// are shared by all workgroups), only the 16 x width activations between layers live in LDS.
// One workgroup = 16 graphs; wave w takes the 16-column output slices w, w + 4, ...
static constexpr int HS_THREADS = 256;
static constexpr int HS_MAXW = 128; // widest hidden layer this form takes

template <int ACT>
__global__ __launch_bounds__(HS_THREADS, 5) void k_head_small(const float *__restrict__ pooled, int B, HeadArgs head,
                                                             float *__restrict__ out, int ldact)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __builtin_amdgcn_s_setprio(3); // (co-runs with the next batch's conv-stack kernel: see k_graph_prep)
    float *sact = reinterpret_cast<float *>(smem); // [2][16][ldact]: ldact = widest hidden layer + 4 (padded rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int g0 = blockIdx.x * 16;
    const int grow = min(g0 + li, B - 1); // (rows past the batch re-read the last graph and are dropped at the store)
    int cur = 0;
#pragma unroll 1
    for (int l = 0; l < head.nlin; l++) {
        const int k = head.dims[l], n = head.dims[l + 1];
        const bool last = (l == head.nlin - 1);
        const float *__restrict__ W = head.w[l];
        const float *__restrict__ bias = head.b[l];
        for (int sl = wave; sl * 16 < n; sl += HS_THREADS / 64) {
            const int nn = sl * 16 + li;
            const int nnc = nn < n ? nn : n - 1;
            const float *wrow = W + (size_t)nnc * k + 4 * lg;
            const float *arow_g = pooled + (size_t)grow * k + 4 * lg; // layer 0: A straight from the pooled matrix
            const float *arow_l = sact + (cur * 16 + li) * ldact + 4 * lg; // later layers: from LDS
            // four accumulator chains over interleaved 16-wide k blocks, 64 k values per step
            f32x4 accs[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                accs[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            auto loadf = [&](int kb, float4 (&a)[4], float4 (&w)[4]) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int kk = kb + 16 * u; // (+ 4 lg inside the row pointers)
                    const bool ok = kk + 4 * lg < k; // k % 4 == 0: a float4 is whole or absent
                    const int kc = ok ? kk : 0;
                    w[u] = *reinterpret_cast<const float4 *>(wrow + kc);
                    a[u] = l == 0 ? *reinterpret_cast<const float4 *>(arow_g + kc)
                                  : *reinterpret_cast<const float4 *>(arow_l + kc);
                    if (!ok)
                        a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            };
            // (no operand double-buffering: the register budget is what lets this kernel share a SIMD with
            // the conv-stack kernel, and it runs in that kernel's shadow anyway)
            for (int kb = 0; kb < k; kb += 64) {
                float4 a[4], w[4];
                loadf(kb, a, w);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, w[u].x, accs[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, w[u].y, accs[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, w[u].z, accs[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, w[u].w, accs[u], 0, 0, 0);
            }
            // C/D: col = lane&15 (output column nn), row = (lane>>4)*4 + r (graph inside the tile)
            if (nn < n) {
                const float bvv = bias ? bias[nn] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gi = lg * 4 + r;
                    const float v = (accs[0][r] + accs[1][r]) + (accs[2][r] + accs[3][r]) + bvv;
                    if (last) {
                        if (g0 + gi < B)
                            out[(size_t)(g0 + gi) * n + nn] = v;
                    } else {
                        sact[((cur ^ 1) * 16 + gi) * ldact + nn] = act_t<ACT>(v);
                    }
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// hipErrorNotSupported when the head's shape does not suit the small form (caller takes k_pool_mlp)
static hipError_t launch_head_small(int num_graphs, const HeadArgs &head, int act, float *out, hipStream_t s,
                                    const float *prepooled)
{
    if (!prepooled || head.nlin < 1 || head.nlin > 8 || (((uintptr_t)prepooled) & 15))
        return hipErrorNotSupported;
    for (int l = 0; l < head.nlin; l++) {
        if ((head.dims[l] & 3) || (((uintptr_t)head.w[l]) & 15))
            return hipErrorNotSupported; // float4 operand fetches
        if (l > 0 && head.dims[l] > HS_MAXW)
            return hipErrorNotSupported; // hidden activations live in the fixed LDS tile
    }
    int maxw = 4;
    for (int l = 1; l < head.nlin; l++)
        maxw = std::max(maxw, (int)head.dims[l]);
    const int ldact = ((maxw + 3) & ~3) + 4;
    const size_t lds = (size_t)2 * 16 * ldact * 4;
    const int grid = (num_graphs + 15) / 16;
    auto go = [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
        hipLaunchKernelGGL(k_head_small<ACT>, dim3(grid), dim3(HS_THREADS), lds, s, prepooled, num_graphs, head, out, ldact);
    };
    GNNB_DISPATCH_ACT(act, go)
    return hipGetLastError();
}

hipError_t launch_pool_mlp(const float *x, const int32_t *node_ptr, int num_graphs, int d,
                           const int32_t *pools, int num_pools, const HeadArgs &head, int act,
                           float *out, hipStream_t s, const float *prepooled)
{
    if (num_graphs <= 0)
        return hipSuccess;
    if (prepooled && options().fuse_head && options().head_small) {
        const hipError_t e = launch_head_small(num_graphs, head, act, out, s, prepooled);
        if (e != hipErrorNotSupported)
            return e;
    }
    const float *src = prepooled ? prepooled : x;
    if (!options().fuse_head || head.nlin < 1 || head.nlin > 8 || (d & 3) || (((uintptr_t)src & 15) != 0))
        return hipErrorNotSupported;
    if (prepooled && (head.dims[0] & 3))
        return hipErrorNotSupported;
    // LDS plan: two activation buffers [16][max width] + every weight matrix
    int maxw0 = 4, maxw1 = 4; // layer l reads buffer l&1 and writes buffer (l+1)&1
    for (int l = 0; l <= head.nlin; l++) {
        if (l & 1)
            maxw1 = std::max(maxw1, head.dims[l]);
        else
            maxw0 = std::max(maxw0, head.dims[l]);
    }
    const int act0_floats = (HEAD_GRAPHS * maxw0 + 3) & ~3, act1_floats = (HEAD_GRAPHS * maxw1 + 3) & ~3;
    int woff[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t wfl = 0;
    for (int l = 0; l < head.nlin; l++) {
        woff[l] = (int)wfl;
        wfl += (((size_t)head.dims[l] * head.dims[l + 1] + 3) & ~(size_t)3) + (((size_t)head.dims[l + 1] + 3) & ~(size_t)3);
        if (((uintptr_t)head.w[l] & 15) != 0)
            return hipErrorNotSupported;
    }
    const size_t lds = ((size_t)act0_floats + act1_floats + wfl) * 4;
    if (lds > 158 * 1024)
        return hipErrorNotSupported; // head too large for the fused kernel: caller uses pool + GEMMs
    int glog2 = 2;
    while ((1 << glog2) < (d >> 2) && glog2 < 6)
        glog2++;
    const int grid = (num_graphs + HEAD_GRAPHS - 1) / HEAD_GRAPHS;
    const int p0 = pools[0], p1 = num_pools > 1 ? pools[1] : 0, p2 = num_pools > 2 ? pools[2] : 0;
    auto go = [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
        auto kern = k_pool_mlp<ACT>;
        (void)ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(HEAD_THREADS), lds, s, x, node_ptr, num_graphs, d, glog2, p0, p1, p2,
                           num_pools, head, out, prepooled, act0_floats, act1_floats, woff[0], woff[1], woff[2], woff[3], woff[4], woff[5],
                           woff[6], woff[7]);
    };
    GNNB_DISPATCH_ACT(act, go)
    return hipGetLastError();
}


// =====================================================================================
// fused 2-layer GCN stack + pooling (the BASELINE config 1 / 2 model family)
// =====================================================================================
// Reference dataflow being fused: compute_gnn_head (the conv layers with skip / activation,
// templates/model.cpp.jinja:151-359; gcn_conv gnn_builder_lib.h:1213-1387, gin_conv :1389-1544) and
// compute_global_graph_pooling (:413-449).  Layer by layer, every intermediate [N, d] matrix makes a
// round trip through HBM (aggregate out -> GEMM in -> GEMM out -> next aggregate in -> ... -> pooling
// in).  A molecule is a few dozen rows, so a handful of WHOLE graphs fit in LDS: here a persistent
// workgroup walks its run of node tiles in stages of <= 64 rows (4 MFMA units) and, per stage,
//   DMA   raw x rows + node records (one buffer, refilled behind P0), dinv + graph boundaries (two buffers) of the
//         NEXT stage -> LDS (global_load_lds)
//   P0    A0 = aggregate(x)                LDS -> LDS   (width F0, eight lanes per row)
//   M0    H  = act(A0 . W0^T + b0)         MFMA 16x16x4, W0 slice in registers -> LDS
//   P1    A1 = aggregate(H)                LDS -> LDS   (lane group per row, padded destination rows)
//   M1    out = act(A1 . W1^T + b1)        MFMA, W1 slice (16 cols x K) in registers; stays in the accumulators
//   PL    pooled[g] = add|mean|max over the rows of each graph of the stage -> HBM
// (stacks of more than two layers and GIN stacks repeat P1 / M inside the stage: the DEEP / GIN variants below).
// HBM traffic = x + tables in, [B, np*d] out: ~14 MB instead of ~270 MB at C2; the kernel is bound by
// the fp32 matrix cores.  Needs: F0 <= 32, h0 in {32,64,128}, h1 <= 128 (h1 % 4 == 0) and the caller's promise
// max_graph_nodes <= 64 - (tile_rows - 1) (validated by graph prep).
// rows per stage: FOUR 16-row MFMA units (64 rows) -- a stage costs ~13 k cycles of barriers and latency chains whatever
// it holds, and three molecules fill 54 of 64 rows where two filled 36 of 48.  The bf16x6 mode keeps three units (its A1
// is three bf16 planes: 1.5x the bytes, and two workgroups must stay resident per CU).
__host__ __device__ constexpr int g2_units(int math) { return math ? 3 : 4; }
static_assert(16 * g2_units(0) == GNNB_G2_STAGE_ROWS && 16 * g2_units(1) == GNNB_G2_STAGE_ROWS_BF6, "graph prep picks the tile size against these");
static constexpr int G2_TCAP = 64;           // tile-table entries a workgroup keeps in LDS
static constexpr int G2_WG = 512;            // 8 waves; two workgroups per CU = 4 waves per SIMD
static constexpr int G2_NW = G2_WG / 64;

struct G2Stage {
    int ta, tb, nb, rows, ga, gb;
};

// Sum / max of a value over the four 16-lane rows of a wave (same lane index in each row) with the
// gfx950 row-swap instructions -- two VALU operations per step instead of an LDS crossbar round trip.
__device__ __forceinline__ float rows4_sum(float x)
{
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows4_max(float x)
{
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// Workgroup barrier of the fused kernel: LDS traffic drained, NO vector-memory drain.  __syncthreads()
// carries a fence, for which the compiler emits s_waitcnt vmcnt(0) whenever it has stores of its own in
// flight (the pooled outputs) -- and that would also wait for the untracked DMA of the next stage.
__device__ __forceinline__ void g2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One MFMA phase of the fused stack: v[k][r] = act(A . Wslice^T + bias) for the wave's 16 columns and
// the rows (rg + k nrg) * 16 + lg * 4 + r of its units k < NU (NU wave-uniform).  The units'
// accumulators are interleaved so that dependent MFMAs are >= 2 issues apart (NU == 1: the k range
// is split over two accumulators instead).
template <int ACT, int KQ, int NU, bool SWZ>
__device__ __forceinline__ void g2_mma(const float *__restrict__ Asrc, int lda, int P, const float (&wr)[KQ * 4],
                                       float bias, int rg, int nrg, int li, int lg, float (&v)[NU][4])
{
    constexpr int NA = NU == 1 ? 2 : NU;
    f32x4 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; a++)
        acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto frag = [&](int q, int k) {
        const int row = (rg + k * nrg) * 16 + li;
        const int c = 4 * q + lg;
        return *reinterpret_cast<const float4 *>(Asrc + row * lda + ((SWZ ? (c ^ (row & (P - 1))) : c) << 2));
    };
    // software pipeline: the fragments of k block q+1 are requested before the MFMAs of block q are issued,
    // so a wave's own LDS latency hides behind its own matrix work
    float4 a4[NU], an[NU];
#pragma unroll
    for (int k = 0; k < NU; k++)
        a4[k] = frag(0, k);
#pragma unroll
    for (int q = 0; q < KQ; q++) {
        if (q + 1 < KQ) {
#pragma unroll
            for (int k = 0; k < NU; k++)
                an[k] = frag(q + 1, k);
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int k = 0; k < NU; k++) {
                const float av = t == 0 ? a4[k].x : (t == 1 ? a4[k].y : (t == 2 ? a4[k].z : a4[k].w));
                const int ai = NU == 1 ? (t & 1) : k;
                acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[q * 4 + t], acc[ai], 0, 0, 0);
            }
        if (q + 1 < KQ) {
#pragma unroll
            for (int k = 0; k < NU; k++)
                a4[k] = an[k];
        }
    }
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            v[k][r] = act_t<ACT>((NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r]) + bias);
}

// M1 of the fused stack with bf16x6: A1 lives in LDS as three bf16 planes [rows][h0] (16-B chunks of
// eight k values, XOR-swizzled by row), the wave's W1 slice as three register sets.  Lane (li, lg) of a
// 16x16x32 MFMA holds k = 32 kb + 8 lg .. + 7 of row / column li for both operands.
template <int ACT, int KB, int NU>
__device__ __forceinline__ void g2_mma_bf6(const char *__restrict__ planes, int plane_bytes, int row_bytes,
                                           const u32x4 (&wh)[KB], const u32x4 (&wm)[KB], const u32x4 (&wl)[KB],
                                           float bias, int li, int lg, float (&v)[NU][4])
{
    f32x4 acc[NU];
#pragma unroll
    for (int k = 0; k < NU; k++)
        acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        bf16x8 ah[NU], am[NU], al[NU];
#pragma unroll
        for (int k = 0; k < NU; k++) {
            const int row = k * 16 + li;
            const int off = row * row_bytes + ((4 * kb + lg) << 4); // rows padded by 16 B: conflict-free without a swizzle
            ah[k] = as_bf16x8(*reinterpret_cast<const u32x4 *>(planes + off));
            am[k] = as_bf16x8(*reinterpret_cast<const u32x4 *>(planes + plane_bytes + off));
            al[k] = as_bf16x8(*reinterpret_cast<const u32x4 *>(planes + 2 * plane_bytes + off));
        }
        const bf16x8 bh = as_bf16x8(wh[kb]), bm = as_bf16x8(wm[kb]), bl = as_bf16x8(wl[kb]);
        // smallest terms first; the units' accumulators interleaved (dependent MFMAs three issues apart)
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[k], bm, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[k], bh, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k], bl, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[k], bh, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k], bm, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k], bh, acc[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            v[k][r] = act_t<ACT>(acc[k][r] + bias);
}

// DEEP: stacks of MORE than two GCN layers (reference compute_gnn_head loops any number of layers,
// model.cpp.jinja:151-359).  The middle layers l = 1 .. nl-2 (width h0 -> h0, skip connection on exactly these,
// models.py:562-564) repeat the P1 / M pair inside the stage with H updated in place; the wave's 16-column weight slice
// is re-read from L2 for every layer and stage (requested in front of P1, consumed behind its barrier) into the
// registers the two-layer form loads once.  The middle layers' weights sit `mid_stride` floats apart (the model blob is
// laid out layer by layer; the launcher checks it).
// GIN (with DEEP): the same stage loop for GIN stacks (reference gin_conv, gnn_builder_lib.h:1389-1544): the aggregate
// is (1 + eps) x_i + sum_j x_j, every layer has TWO linears (ReLU between them, the model's activation and the skip
// connection behind the second), all wide matrices hidden x hidden at one stride in the blob: index 0 = layer 0's
// second linear, 2l - 1 / 2l = layer l's first / second.  A linear whose input and output share a buffer multiplies,
// waits for everybody at a barrier, then writes.
template <int ACT, int KQ0, int KQ1, int MATH, bool DEEP = false, bool GIN = false>
__global__ __launch_bounds__(G2_WG, 4) void k_gcn2_fused(
    const float *__restrict__ x, int f0, const int4 *__restrict__ node_rec,
    const int32_t *__restrict__ col, const float *__restrict__ dinv,
    const int32_t *__restrict__ tile_first, const int32_t *__restrict__ tile_graph,
    const int32_t *__restrict__ node_ptr, int num_tiles, int num_graphs, int N, const float *__restrict__ W0,
    const float *__restrict__ b0, int h0, const float *__restrict__ W1, const float *__restrict__ b1,
    int h1, int p0, int p1, int p2, int np, float *__restrict__ pooled, int nl,
    const float *__restrict__ Wmid, const float *__restrict__ bmid, long mid_stride, long bmid_stride, int skip,
    float gin_eps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int G2_UNITS = g2_units(MATH), G2_CAP = 16 * G2_UNITS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // ---- LDS carve (bytes, every region 16-B aligned):
    //   rows  xs | srec                ONE buffer: read by P0 only, refilled right behind P0
    //   small sdinv | node_ptr of <= 64 graphs (+ end)   TWO buffers (P1 and the pooling still read them)
    //   H | A1 (A0 lives in the head of A1: P0 writes it, M0 reads it, P1 overwrites it) | REC | tile tables
    const int xs_b = ((G2_CAP * f0 * 4) + 15) & ~15;
    const int rows_b = xs_b + G2_CAP * 32;
    const int small_b = G2_CAP * 4 + 272;
    const int ldh = (h0 > h1 ? h0 : h1) + 4;          // padded H row (floats)
    // NOTE: LDS pointers are always derived arithmetically from `smem`.  Indexing an array of LDS
    // pointers with a runtime value makes the compiler lose the address space and emit FLAT loads,
    // whose s_waitcnt vmcnt(0) also waits for the in-flight DMA of the next stage.
    constexpr int LD0 = 16 * KQ0; // A0 row: F0 values zero-padded to whole 16-wide MFMA k blocks
    float *H = reinterpret_cast<float *>(smem + rows_b + 2 * small_b);
    float *A1 = H + G2_CAP * ldh;
    float *A0 = A1;
    // per-row aggregation record written by P0, read by P1: {byte offsets of the 4 inline neighbour rows in H}
    // {coefficients dinv_i dinv_j, 0 past the degree} {dinv_i^2, rp0, deg, dinv_i}
    // (MATH 1: A1 is three bf16 planes [G2_CAP][h0] instead of one fp32 matrix: 1.5x the bytes)
    constexpr int KB1 = KQ1 / 2 > 0 ? KQ1 / 2 : 1; // 32-wide k blocks of layer 1 (h0 = 32, 64, 128)
    // A1 rows are padded (fp32: +4 floats, bf16 planes: +16 B) instead of XOR-swizzled: M1's fragment reads
    // (8 lanes x 16 B per cycle, consecutive rows) then fall into distinct bank groups AND their addresses are
    // base + immediate -- the swizzle cost two VALU operations per read, and VALU issue is what this kernel
    // runs out of
    const int lda1 = h0 + 4, prow_b = h0 * 2 + 16;
    const int plane_b = G2_CAP * prow_b;
    int4 *REC = reinterpret_cast<int4 *>(reinterpret_cast<char *>(A1) + (MATH ? 3 * plane_b : G2_CAP * lda1 * 4));
    int32_t *stile = reinterpret_cast<int32_t *>(REC + 3 * G2_CAP);
    int32_t *sgraph = stile + (G2_TCAP + 1);

    const int t0 = (int)(((long long)blockIdx.x * num_tiles) / gridDim.x);
    const int t1 = (int)(((long long)(blockIdx.x + 1) * num_tiles) / gridDim.x);
    if (t1 <= t0)
        return;
    // (clamped: the tables of a malformed batch may hold stale entries; a flagged batch must still stay in range)
    for (int i = tid; i <= t1 - t0; i += G2_WG) {
        stile[i] = min(max(tile_first[t0 + i], 0), N);
        sgraph[i] = min(max(tile_graph[t0 + i], 0), num_graphs);
    }
    __syncthreads();

    auto plan = [&](int ta) {
        G2Stage st;
        st.ta = ta;
        st.tb = ta;
        st.nb = 0;
        st.rows = 0;
        st.ga = 0;
        st.gb = 0;
        if (ta >= t1)
            return st;
        st.nb = stile[ta - t0];
        int tb = ta + 1;
        while (tb < t1 && stile[tb + 1 - t0] - st.nb <= G2_CAP)
            tb++;
        st.tb = tb;
        st.rows = max(min(stile[tb - t0] - st.nb, G2_CAP), 0); // (> CAP only if the max_graph_nodes promise is broken)
        st.ga = sgraph[ta - t0];
        // (empty graphs after the last node belong to the last stage: when N is a multiple of the tile
        // size the first of them already owns tile_graph[num_tiles])
        st.gb = max(tb == num_tiles ? num_graphs : sgraph[tb - t0], st.ga);
        return st;
    };
    // the stage's rows (x, node records) -> the single rows buffer: behind P0 of the stage before
    auto issue_rows = [&](const G2Stage &st, int lane, int wave) { // (lane, wave: see `tv` below)
        if (st.ta >= t1)
            return;
        dma_dwords_u(x + (size_t)st.nb * f0, smem, st.rows * f0, wave, lane, G2_NW);
        const char *grec = reinterpret_cast<const char *>(node_rec + 2 * (size_t)st.nb);
        const int rbytes = st.rows * 32;
        if (wave * 1024 + lane * 16 < rbytes) // <= 64 rows * 32 B = 2 KiB: waves 0 and 1
            dma16_to_lds_u(grec + wave * 1024 + lane * 16, smem + xs_b + wave * 1024);
    };
    // its normalisers and graph boundaries -> small buffer bb: at the top of the stage before
    auto issue_small = [&](const G2Stage &st, int bb, int lane, int wave) {
        if (st.ta >= t1)
            return;
        // (a stage may have NO rows and still own graphs: empty graphs behind a graph that ends on the
        // tile edge -- their boundaries are still needed by the pooling phase)
        char *base = smem + rows_b + (size_t)bb * small_b;
        if (wave == 2 && lane < st.rows)
            dma4_to_lds_u(dinv + st.nb + lane, base);
        // graph boundaries of the stage for the pooling phase (first 64 graphs; more only if empty
        // graphs pile up, those are read from global memory)
        const int ng = min(st.gb - st.ga, 64) + 1;
        if (wave == 3 && lane < ng)
            dma4_to_lds_u(node_ptr + st.ga + lane, base + G2_CAP * 4);
        if (wave == 4 && lane + 64 < ng)
            dma4_to_lds_u(node_ptr + st.ga + 64 + lane, base + G2_CAP * 4 + 256);
    };

    // the first stage's inputs start their way to LDS before the weights are fetched (both are waited for
    // together below), instead of after them
    G2Stage cur = plan(t0);
    issue_small(cur, 0, lane, wave);
    issue_rows(cur, lane, wave);


    // ---- wave roles: layer L has ncs_L = pow2ceil(h_L / 16) column slices of 16 and nrg_L = 8 / ncs_L
    // row groups; wave w owns slice (w mod ncs) for the units rg, rg + nrg, ... with rg = w / ncs
    int cs0l = 0, cs1l = 0;
    while ((16 << cs0l) < h0)
        cs0l++;
    while ((16 << cs1l) < h1)
        cs1l++; // h1 <= 128 -> <= 3
    const int nrg0 = G2_NW >> cs0l, n0c = (wave & ((1 << cs0l) - 1)) * 16 + li;
    const int n1c = (wave & ((1 << cs1l) - 1)) * 16 + li;

    // ---- weight slices -> registers (16 output columns x K per layer and wave)
    float w0r[KQ0 * 4], w1r[KQ1 * 4];
#pragma unroll
    for (int q = 0; q < KQ0; q++) {
        const int k = 16 * q + 4 * lg;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0c < h0)
            v = load4_guard(W0 + (size_t)n0c * f0 + k, f0 - k, false);
        w0r[q * 4 + 0] = v.x;
        w0r[q * 4 + 1] = v.y;
        w0r[q * 4 + 2] = v.z;
        w0r[q * 4 + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < KQ1; q++) {
        const int k = 16 * q + 4 * lg; // h0 == 16 * KQ1
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!MATH && !DEEP && n1c < h1)
            v = *reinterpret_cast<const float4 *>(W1 + (size_t)n1c * h0 + k);
        w1r[q * 4 + 0] = v.x;
        w1r[q * 4 + 1] = v.y;
        w1r[q * 4 + 2] = v.z;
        w1r[q * 4 + 3] = v.w;
    }
    float bias0 = (n0c < h0 && b0) ? b0[n0c] : 0.0f;
    float bias1 = (n1c < h1 && b1) ? b1[n1c] : 0.0f;
    // MATH 1: the wave's W1 slice as three bf16 register sets, lane (li, lg) holding k = 32 kb + 8 lg .. + 7
    u32x4 wh[KB1], wm[KB1], wl[KB1];
    if (MATH) {
#pragma unroll
        for (int kb = 0; kb < KB1; kb++) {
            float wv[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
                wv[i] = 0.0f;
            if (n1c < h1 && 32 * kb + 8 * lg < h0) {
                const float4 v0 = *reinterpret_cast<const float4 *>(W1 + (size_t)n1c * h0 + 32 * kb + 8 * lg);
                const float4 v1 = *reinterpret_cast<const float4 *>(W1 + (size_t)n1c * h0 + 32 * kb + 8 * lg + 4);
                wv[0] = v0.x, wv[1] = v0.y, wv[2] = v0.z, wv[3] = v0.w;
                wv[4] = v1.x, wv[5] = v1.y, wv[6] = v1.z, wv[7] = v1.w;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t h0_, m0_, l0_, h1_, m1_, l1_;
                split3(wv[2 * i], h0_, m0_, l0_);
                split3(wv[2 * i + 1], h1_, m1_, l1_);
                wh[kb][i] = pack_hi16(h0_, h1_);
                wm[kb][i] = pack_hi16(m0_, m1_);
                wl[kb][i] = pack_hi16(l0_, l1_);
            }
        }
    }
    // Pin every weight register through an (empty) asm: the compiler must finish the loads HERE.  Left
    // alone it keeps them "possibly in flight" around the stage loop's back edge and guards their first
    // use in M0 / M1 with s_waitcnt vmcnt(0) -- which also waits for the next stage's DMA issued just
    // before, i.e. exposes the full memory latency in every stage.
#pragma unroll
    for (int q = 0; q < KQ0 * 4; q++)
        asm volatile("" : "+v"(w0r[q]));
    if (MATH) {
#pragma unroll
        for (int kb = 0; kb < KB1; kb++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                asm volatile("" : "+v"(wh[kb][i]), "+v"(wm[kb][i]), "+v"(wl[kb][i]));
    } else {
#pragma unroll
        for (int q = 0; q < KQ1 * 4; q++)
            asm volatile("" : "+v"(w1r[q]));
    }
    asm volatile("" : "+v"(bias0), "+v"(bias1));
    __syncthreads();

    const int nv1 = h0 >> 2;                         // float4 chunks per H row consumed by layer 1
    int glog2 = 2;
    while ((1 << glog2) < nv1 && glog2 < 6)
        glog2++;
    const int Gl = 1 << glog2, groups = G2_WG >> glog2;
    const int pools[3] = {p0, p1, p2};

#ifdef GNNB_PROBE
    unsigned long long pt[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = clock64(), pw0 = wall_clock64(), pt_last = pt0;
    int nst = 0;
#define G2_PT(i) do { const unsigned long long _n = clock64(); pt[i] += _n - pt_last; pt_last = _n; } while (0)
#else
#define G2_PT(i) do { } while (0)
#endif
    int b = 0;
    int stores_behind_dma = 0; // wave-uniform: store INSTRUCTIONS this wave issued since its last DMA issue
    while (cur.ta < t1) {
        const G2Stage nxt = plan(cur.tb);
        // Stage `cur` has landed (untracked DMA: the wait is ours).  Vector-memory operations retire in
        // order and the previous stage's pooled stores were issued AFTER this DMA, so waiting for
        // "at most <that many> outstanding" proves the DMA done and leaves the stores in flight.
        vmcnt_wait_upto(stores_behind_dma);
        stores_behind_dma = 0;
        g2_barrier(); // (1) everyone's DMA is in; everyone is done with the previous stage
        G2_PT(0);
        const char *sbase = smem + rows_b + (size_t)b * small_b;
        const float *xs = reinterpret_cast<const float *>(smem);
        const int4 *srec = reinterpret_cast<const int4 *>(smem + xs_b);
        const float *sdinv = reinterpret_cast<const float *>(sbase);
        const int32_t *sgp = reinterpret_cast<const int32_t *>(sbase + G2_CAP * 4);
        const int rows = cur.rows, nb = cur.nb;
        const int units = (rows + 15) >> 4;
        // The thread index is re-made OPAQUE every stage and every per-lane quantity below is derived from
        // it again (a dozen VALU ops).  Otherwise the compiler hoists ~50 loop-invariant LDS offsets out of
        // the stage loop, runs out of its 128 registers and parks them in scratch -- whose reloads are
        // vector-memory operations that queue behind the next stage's DMA.
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int li = tv & 15, lg = (tv >> 4) & 3, wv = tv >> 6;
        const int n0c = (wv & ((1 << cs0l) - 1)) * 16 + li, n1c = (wv & ((1 << cs1l) - 1)) * 16 + li;
        const int rg0 = wv >> cs0l;
        const int grp = tv >> glog2, gl = tv & (Gl - 1);
        issue_small(nxt, b ^ 1, tv & 63, wv);
        G2_PT(1);

        // ---- P0: A0[i][f] = sum_j x_j[f] dinv_i dinv_j + x_i[f] dinv_i^2   (CSR order, self last)
        // Eight lanes per row, lane l8 takes features l8, l8 + 8, ...: all <= 64 rows in ONE pass of the 512
        // threads (lanes f >= F0 write the zero padding).  Every LDS load is unconditional -- unused neighbour
        // slots alias the row itself, inactive threads read row 0 -- and the degree only selects: a
        // lane-divergent guard around a load makes the compiler wait at every join.
        {
            constexpr int T0 = LD0 / 8;
            const int i = tv >> 3, l8 = tv & 7;
            const bool active = i < rows;
            const int ic = active ? i : 0;
            const int4 r0 = srec[2 * ic], r1 = srec[2 * ic + 1];
            const int deg = r0.y;
            const int jl[4] = {r0.z - nb, r0.w - nb, r1.x - nb, r1.y - nb};
            const float di = GIN ? 1.0f : sdinv[ic];
            float xv[T0][4], xself[T0], sv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                sv[q] = GIN ? 1.0f : sdinv[jl[q]];
#pragma unroll
                for (int t = 0; t < T0; t++) {
                    const int f = l8 + 8 * t;
                    xv[t][q] = xs[jl[q] * f0 + (f < f0 ? f : 0)];
                }
            }
#pragma unroll
            for (int t = 0; t < T0; t++) {
                const int f = l8 + 8 * t;
                xself[t] = xs[ic * f0 + (f < f0 ? f : 0)];
            }
            float c[4], acc[T0];
#pragma unroll
            for (int q = 0; q < 4; q++)
                c[q] = deg > q ? di * sv[q] : 0.0f;
#pragma unroll
            for (int t = 0; t < T0; t++) {
                acc[t] = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    acc[t] += xv[t][q] * c[q];
            }
            if (active) {
                // (degree > 4: rare in molecules.  The tracked global read makes the compiler drain vmcnt, i.e. the wave
                // also waits for the next stage's DMA; measured bound of staging the CSR slice in LDS instead: the kernel
                // without this loop altogether is 55.2 vs 57.0 us)
                for (int k = r0.x + 4; k < r0.x + deg; k++) {
                    const int j = col[k] - nb;
                    const float cj = GIN ? 1.0f : di * sdinv[j];
#pragma unroll
                    for (int t = 0; t < T0; t++) {
                        const int f = l8 + 8 * t;
                        acc[t] += xs[j * f0 + (f < f0 ? f : 0)] * cj;
                    }
                }
#pragma unroll
                for (int t = 0; t < T0; t++) {
                    const int f = l8 + 8 * t;
                    A0[i * LD0 + f] = f < f0 ? acc[t] + xself[t] * (GIN ? 1.0f + gin_eps : di * di) : 0.0f;
                }
                if (l8 == 0) { // the row's scalars, computed once here instead of by every lane of P1's lane group
                    REC[3 * i] = make_int4(jl[0] * ldh * 4, jl[1] * ldh * 4, jl[2] * ldh * 4, jl[3] * ldh * 4);
                    REC[3 * i + 1] = make_int4(__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]), __float_as_int(c[3]));
                    REC[3 * i + 2] = make_int4(__float_as_int(GIN ? 1.0f + gin_eps : di * di), r0.x, deg, __float_as_int(di));
                }
            }
        }
        G2_PT(2);
        g2_barrier(); // (2)
        issue_rows(nxt, tv & 63, wv); // (P0 was the last reader of the rows buffer)
        G2_PT(3);

        // (DEEP) this wave's weight slice + bias for a 128-wide layer -> the w1r registers: ordinary loads, requested
        // here, first used behind the next barrier
        auto load_slice = [&](const float *Wl, const float *bl, int ncol, int nlim) {
#pragma unroll
            for (int q = 0; q < KQ1; q++) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ncol < nlim)
                    v = *reinterpret_cast<const float4 *>(Wl + (size_t)ncol * h0 + 16 * q + 4 * lg);
                w1r[q * 4 + 0] = v.x;
                w1r[q * 4 + 1] = v.y;
                w1r[q * 4 + 2] = v.z;
                w1r[q * 4 + 3] = v.w;
            }
            bias1 = (ncol < nlim && bl) ? bl[ncol] : 0.0f;
        };
        if (GIN)
            load_slice(Wmid, bmid, n0c, h0); // layer 0's second linear: in flight behind M0

        // ---- M0: H = act(A0 . W0^T + b0)   (wave: column slice x row group)
        {
            const int nu = rg0 < units ? (units - rg0 + nrg0 - 1) / nrg0 : 0;
            auto m0 = [&](auto nutag) {
                constexpr int NU = decltype(nutag)::value;
                float v[NU][4];
                g2_mma<GIN ? (int)GNNB_ACT_RELU : ACT, KQ0, NU, false>(A0, LD0, 1, w0r, bias0, rg0, nrg0, li, lg, v);
                if (n0c < h0) {
#pragma unroll
                    for (int k = 0; k < NU; k++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            H[((rg0 + k * nrg0) * 16 + lg * 4 + r) * ldh + n0c] = v[k][r];
                }
            };
            if (G2_UNITS > 3 && nu == 4)
                m0(IntTag<G2_UNITS>{});
            else if (nu == 3)
                m0(IntTag<3>{});
            else if (nu == 2)
                m0(IntTag<2>{});
            else if (nu == 1)
                m0(IntTag<1>{});
        }
        G2_PT(4);
        g2_barrier(); // (3)
        G2_PT(5);

        // ---- P1: A1 = gcn-aggregate(H), one lane group of h0/4 lanes per row (a float4 chunk each),
        // destination rows padded for M1's fragment reads.  Offsets and coefficients come ready-made from
        // REC; the next pass's record is fetched while this pass's rows are in flight.  (VALU instructions
        // are what bounds this kernel: row-level scalars must not be recomputed by all lanes of a group.)
        auto phase_p1 = [&]() {
            typedef Vf<4> V;
            const char *Hl = reinterpret_cast<const char *>(H) + gl * 16; // this lane's chunk of row 0
            int4 ra = make_int4(0, 0, 0, 0), rc = ra, rd = ra;
            if (grp < rows) {
                ra = REC[3 * grp];
                rc = REC[3 * grp + 1];
                rd = REC[3 * grp + 2];
            }
            // (at most G2_UNITS passes: groups >= 16; fixed-count loop, no derived trip count)
#pragma unroll 1
            for (int pass = 0; pass < G2_UNITS; pass++) {
                const int rA = grp + pass * groups;
                if (rA >= rows)
                    break;
                const int rN = rA + groups;
                const int4 ja = ra, ca = rc, da = rd;
                if (rN < rows) {
                    ra = REC[3 * rN];
                    rc = REC[3 * rN + 1];
                    rd = REC[3 * rN + 2];
                }
                const V n0 = V::load(reinterpret_cast<const float *>(Hl + ja.x)); // unused slots alias the row itself (coefficient 0)
                const V n1 = V::load(reinterpret_cast<const float *>(Hl + ja.y));
                const V n2 = V::load(reinterpret_cast<const float *>(Hl + ja.z));
                const V n3 = V::load(reinterpret_cast<const float *>(Hl + ja.w));
                const V selfA = V::load(reinterpret_cast<const float *>(Hl + rA * ldh * 4));
                V accA = vmul(n0, V::splat(__int_as_float(ca.x)));
                accA = vadd(accA, vmul(n1, V::splat(__int_as_float(ca.y))));
                accA = vadd(accA, vmul(n2, V::splat(__int_as_float(ca.z))));
                accA = vadd(accA, vmul(n3, V::splat(__int_as_float(ca.w))));
                for (int k = da.y + 4; k < da.y + da.z; k++) { // degree > 4
                    const int j = col[k] - nb;
                    accA = vadd(accA, vmul(V::load(reinterpret_cast<const float *>(Hl + j * ldh * 4)),
                                           V::splat(GIN ? 1.0f : __int_as_float(da.w) * sdinv[j])));
                }
                accA = vadd(accA, vmul(selfA, V::splat(__int_as_float(da.x))));
                if (MATH) {
                    // split into the three bf16 planes: this lane's 4 values = half of a 16-B chunk of 8
                    uint32_t hh[4], mm[4], ll[4];
                    split3(accA.v.x, hh[0], mm[0], ll[0]);
                    split3(accA.v.y, hh[1], mm[1], ll[1]);
                    split3(accA.v.z, hh[2], mm[2], ll[2]);
                    split3(accA.v.w, hh[3], mm[3], ll[3]);
                    char *dstp = reinterpret_cast<char *>(A1) + rA * prow_b + gl * 8;
                    *reinterpret_cast<uint2 *>(dstp) = make_uint2(pack_hi16(hh[0], hh[1]), pack_hi16(hh[2], hh[3]));
                    *reinterpret_cast<uint2 *>(dstp + plane_b) = make_uint2(pack_hi16(mm[0], mm[1]), pack_hi16(mm[2], mm[3]));
                    *reinterpret_cast<uint2 *>(dstp + 2 * plane_b) = make_uint2(pack_hi16(ll[0], ll[1]), pack_hi16(ll[2], ll[3]));
                } else {
                    accA.store(A1 + rA * lda1 + gl * 4);
                }
            }
        };
        // ---- M (a 128-wide layer whose output replaces H): H = act(A1 . Wl^T + bl (+ H)) -- a lane reads exactly the
        // elements it writes, so the skip term needs no second buffer
        auto m_mid = [&]() {
            const int nu = rg0 < units ? (units - rg0 + nrg0 - 1) / nrg0 : 0;
            auto mm = [&](auto nutag) {
                constexpr int NU = decltype(nutag)::value;
                float v[NU][4];
                g2_mma<GNNB_ACT_NONE, KQ1, NU, false>(A1, lda1, 1, w1r, bias1, rg0, nrg0, li, lg, v);
                if (n0c < h0) {
#pragma unroll
                    for (int k = 0; k < NU; k++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float *hp = H + ((rg0 + k * nrg0) * 16 + lg * 4 + r) * ldh + n0c;
                            *hp = act_t<ACT>(v[k][r] + (skip ? *hp : 0.0f));
                        }
                }
            };
            if (G2_UNITS > 3 && nu == 4)
                mm(IntTag<G2_UNITS>{});
            else if (nu == 3)
                mm(IntTag<3>{});
            else if (nu == 2)
                mm(IntTag<2>{});
            else if (nu == 1)
                mm(IntTag<1>{});
        };
        // ---- M in place (GIN: input and output share `buf`): multiply, barrier (everybody has read), write, barrier
        auto m_inplace = [&](float *buf, int ld, auto acttag, int next_wide) {
            constexpr int A = decltype(acttag)::value;
            const int nu = rg0 < units ? (units - rg0 + nrg0 - 1) / nrg0 : 0;
            float v[G2_UNITS][4];
            auto comp = [&](auto nutag) {
                constexpr int NU = decltype(nutag)::value;
                float t[NU][4];
                g2_mma<A, KQ1, NU, false>(buf, ld, 1, w1r, bias1, rg0, nrg0, li, lg, t);
#pragma unroll
                for (int k = 0; k < NU; k++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        v[k][r] = t[k][r];
            };
            if (G2_UNITS > 3 && nu == 4)
                comp(IntTag<G2_UNITS>{});
            else if (nu == 3)
                comp(IntTag<3>{});
            else if (nu == 2)
                comp(IntTag<2>{});
            else if (nu == 1)
                comp(IntTag<1>{});
            // the weight registers are free again: request the next linear's slice now, its latency hides behind the
            // two barriers and the write-back
            if (next_wide >= 0)
                load_slice(Wmid + (size_t)next_wide * mid_stride, bmid + (size_t)next_wide * bmid_stride, n0c, h0);
            g2_barrier();
            if (n0c < h0) {
#pragma unroll
                for (int k = 0; k < G2_UNITS; k++)
                    if (k < nu) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            buf[((rg0 + k * nrg0) * 16 + lg * 4 + r) * ld + n0c] = v[k][r];
                    }
            }
            g2_barrier();
        };
        if (GIN) {
            // layer 0's second linear, then per further layer: aggregate, first linear (ReLU, in place on A1), second
            // linear (-> H with skip + activation; the LAST one stays in the accumulators for the pooling below)
            m_inplace(H, ldh, IntTag<ACT>{}, 1); // (its own slice, index 0, was requested in front of M0)
            for (int l = 1; l < nl; l++) {
                // (layer l's first slice, index 2l - 1, is in flight since the previous in-place product)
                phase_p1();
                g2_barrier();
                m_inplace(A1, lda1, IntTag<GNNB_ACT_RELU>{}, 2 * l);
                if (l + 1 < nl) {
                    m_mid();
                    load_slice(Wmid + (size_t)(2 * l + 1) * mid_stride, bmid + (size_t)(2 * l + 1) * bmid_stride, n0c, h0);
                    g2_barrier();
                }
            }
        } else {
            if (DEEP) {
                for (int l = 1; l + 1 < nl; l++) {
                    load_slice(Wmid + (size_t)(l - 1) * mid_stride, bmid ? bmid + (size_t)(l - 1) * bmid_stride : nullptr, n0c, h0);
                    __builtin_amdgcn_sched_barrier(0); // (keep the requests in FRONT of P1: their latency hides behind it)
                    phase_p1();
                    g2_barrier();
                    m_mid();
                    g2_barrier();
                }
                load_slice(W1, b1, n1c, h1);
                __builtin_amdgcn_sched_barrier(0);
            }
            phase_p1();
        }
        G2_PT(6);
        g2_barrier(); // (4)
        G2_PT(7);

        // ---- M1 + pooling: out = act(A1 . W1^T + b1) stays in the accumulators (the wave owns its 16
        // columns for ALL rows of the stage; waves beyond h1/16 slices idle) and is pooled per graph in
        // registers: masked add / max over the lane's 4 rows per unit, then across the four 16-lane groups.
        // (reference global_add/mean/max_pool, gnn_builder_lib.h:2709-2803; rows in order within a lane,
        // lane groups combined pairwise)
        if (wv < (1 << cs1l) && units > 0) {
            auto m1 = [&](auto nutag) {
                constexpr int NU = decltype(nutag)::value;
                float v[NU][4];
                if (MATH)
                    g2_mma_bf6<ACT, KB1, NU>(reinterpret_cast<const char *>(A1), plane_b, prow_b, wh, wm, wl, bias1, li, lg, v);
                else
                    g2_mma<ACT, KQ1, NU, false>(A1, lda1, 1, w1r, bias1, 0, 1, li, lg, v);
                const int ngr = cur.gb - cur.ga;
                // (one store instruction per graph and pool; none if the whole slice is past h1; the rare
                // paths below that read global memory only make the count conservative -- see the wait)
                stores_behind_dma = wv * 16 < h1 ? ngr * np : 0;
                auto pool_graph = [&](int gi, int r0g, int r1g) { // wave-uniform row range of graph ga + gi
                    r0g = __builtin_amdgcn_readfirstlane(r0g) - nb;
                    r1g = min(__builtin_amdgcn_readfirstlane(r1g) - nb, G2_CAP);
                    float sum = 0.0f, mx = -INFINITY;
#pragma unroll
                    for (int k = 0; k < NU; k++) {
                        if (r1g > k * 16 && r0g < k * 16 + 16) { // uniform: the unit overlaps the graph
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int row = k * 16 + lg * 4 + r;
                                const bool in = row >= r0g && row < r1g;
                                sum += in ? v[k][r] : 0.0f;
                                mx = fmaxf(mx, in ? v[k][r] : -INFINITY);
                            }
                        }
                    }
                    sum = rows4_sum(sum);
                    mx = rows4_max(mx);
                    const int n = r1g - r0g;
                    if (lg == 0 && n1c < h1) {
#pragma unroll
                        for (int kk = 0; kk < 3; kk++) {
                            if (kk >= np)
                                break;
                            float rr = sum;
                            if (pools[kk] == GNNB_POOL_MEAN)
                                rr = n > 0 ? sum / (float)n : 0.0f;
                            else if (pools[kk] == GNNB_POOL_MAX)
                                rr = n > 0 ? mx : 0.0f;
                            pooled[((size_t)(cur.ga + gi) * np + kk) * h1 + n1c] = rr;
                        }
                    }
                };
                // two loops, not one with a choice inside: a select between the LDS table and global
                // memory is if-converted into flat loads (+ a full vmcnt/lgkmcnt drain per graph)
                const int nlds = min(ngr, 64);
                for (int gi = 0; gi < nlds; gi++)
                    pool_graph(gi, sgp[gi], sgp[gi + 1]);
                for (int gi = nlds; gi < ngr; gi++) // a pile of empty graphs
                    pool_graph(gi, node_ptr[cur.ga + gi], node_ptr[cur.ga + gi + 1]);
            };
            if (G2_UNITS > 3 && units == 4)
                m1(IntTag<G2_UNITS>{});
            else if (units == 3)
                m1(IntTag<3>{});
            else if (units == 2)
                m1(IntTag<2>{});
            else
                m1(IntTag<1>{});
        } else if (units == 0 && wv == 0) {
            // a stage without rows (empty graphs behind the last node of a tile): zeros
            stores_behind_dma = 1 << 20; // (full drain)
            for (int e = tv; e < (cur.gb - cur.ga) * np * h1; e += 64)
                pooled[(size_t)cur.ga * np * h1 + e] = 0.0f;
        }
        G2_PT(10);
#ifdef GNNB_PROBE
        nst++;
#endif
        cur = nxt;
        b ^= 1;
    }
#ifdef GNNB_PROBE
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long *o = g_probe + 8 * 8192 + (blockIdx.x * 8 + wave) * 16; // second half: other kernels stamp the first
        o[0] = pw0;
        o[1] = wall_clock64();
        for (int i = 0; i < 11; i++)
            o[2 + i] = pt[i];
        o[13] = clock64() - pt0;
        o[14] = (unsigned long long)nst;
    }
#endif
}

// node tiles the fused stack can walk in one launch: every resident workgroup keeps its run of the tile table in LDS
// (graph prep coarsens the tiles of very large batches against this, so that they stay on the fused path)
long gcn2_fused_tile_capacity()
{
    int devid = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess)
        cus = prop.multiProcessorCount;
    return (long)(G2_TCAP - 2) * 2 * cus;
}

hipError_t launch_gcn2_fused(const BatchTables &t, const float *x, int f0, const float *w0, const float *b0,
                             int h0, const float *w1, const float *b1, int h1, int act,
                             const int32_t *pools, int num_pools, float *pooled, hipStream_t s, const G2Deep &deep)
{
    const Options &o = options();
    if (!o.fuse_gcn2 || t.num_nodes <= 0)
        return hipErrorNotSupported;
    // more than two layers: fp32 mode only, middle weights 16-B aligned (float4 slice loads)
    if (deep.nl < 2 || (deep.nl > 2 && (o.math || !deep.wmid || (((uintptr_t)deep.wmid) & 15) || (deep.mid_stride & 3))))
        return hipErrorNotSupported;
    // GIN stacks: fp32 mode, hidden == out (every wide matrix h0 x h0), biases present
    if (deep.gin && (o.math || h1 != h0 || !deep.wmid || !deep.bmid || (((uintptr_t)deep.wmid) & 15) || (deep.mid_stride & 3)))
        return hipErrorNotSupported;
    const int math = o.math ? 1 : 0;
    const int cap = 16 * g2_units(math);
    if (t.max_graph_nodes_hint <= 0 || t.max_graph_nodes_hint + t.tile_rows - 1 > cap)
        return hipErrorNotSupported; // no promise that whole graphs fit a stage
    if (f0 < 1 || f0 > 32 || !(h0 == 32 || h0 == 64 || h0 == 128) || h1 < 4 || h1 > 128 || (h1 & 3))
        return hipErrorNotSupported;
    if ((((uintptr_t)w1) & 15) || (((uintptr_t)pooled) & 15) || (((uintptr_t)x) & 3))
        return hipErrorNotSupported;
    // every stage must hold at least one tile: workgroups need ceil(T / grid) + 1 <= G2_TCAP table entries
    // (LDS carve: see the kernel)
    const int xs_b = ((cap * f0 * 4) + 15) & ~15;
    const int rows_b = xs_b + cap * 32, small_b = cap * 4 + 272;
    const int ldh = (h0 > h1 ? h0 : h1) + 4;
    const size_t a1_b = std::max((size_t)cap * (math ? 3 * (h0 * 2 + 16) : (h0 + 4) * 4), (size_t)cap * 16 * (f0 <= 16 ? 1 : 2) * 4);
    const size_t lds = (size_t)rows_b + 2 * (size_t)small_b + (size_t)cap * ldh * 4 + a1_b + (size_t)cap * 48 +
                       2 * (size_t)(G2_TCAP + 1) * 4;
    const int kq0 = f0 <= 16 ? 1 : 2, kq1 = h0 / 16;
    const int p0 = pools[0], p1 = num_pools > 1 ? pools[1] : 0, p2 = num_pools > 2 ? pools[2] : 0;
    hipError_t rc = hipErrorNotSupported;
    auto go2 = [&](auto atag, auto q0tag, auto q1tag, auto mtag, auto dtag) {
        constexpr int ACT = decltype(atag)::value, KQ0 = decltype(q0tag)::value, KQ1 = decltype(q1tag)::value;
        constexpr int MATH = decltype(mtag)::value;
        constexpr bool DEEP = decltype(dtag)::value != 0, GIN = decltype(dtag)::value == 2;
        auto kern = k_gcn2_fused<ACT, KQ0, KQ1, MATH, DEEP, GIN>;
        if (ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds) != hipSuccess) {
            rc = hipErrorNotSupported;
            return;
        }
        static size_t lds_set = 0; // (occupancy of this instantiation at this LDS size: the same on every MI355X of a node)
        static int blocks = 0, cus = 256;
        if (lds_set != lds) {
            int nb = 0, devid = 0;
            hipDeviceProp_t prop;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, G2_WG, lds) != hipSuccess || nb < 1)
                nb = 1;
            if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess)
                cus = prop.multiProcessorCount;
            blocks = nb > 2 ? 2 : nb;
            lds_set = lds;
        }
        long long grid = (long long)cus * blocks;
        if (grid > t.num_tiles)
            grid = t.num_tiles;
        const long long min_grid = ((long long)t.num_tiles + G2_TCAP - 2) / (G2_TCAP - 1);
        if (grid < min_grid) {
            rc = hipErrorNotSupported;
            return;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G2_WG), lds, s, x, f0, t.node_rec, t.col, t.dinv,
                           t.tile_first, t.tile_graph, t.graph_ptr, t.num_tiles, t.num_graphs, t.num_nodes, w0, b0, h0, w1, b1, h1, p0, p1, p2,
                           num_pools, pooled, deep.nl, deep.wmid, deep.bmid, deep.mid_stride, deep.bmid_stride, deep.skip, deep.eps);
        rc = hipGetLastError();
    };
    auto go = [&](auto atag, auto q0tag, auto q1tag) {
        if (deep.gin)
            go2(atag, q0tag, q1tag, IntTag<0>{}, IntTag<2>{});
        else if (math)
            go2(atag, q0tag, q1tag, IntTag<1>{}, IntTag<0>{});
        else if (deep.nl > 2)
            go2(atag, q0tag, q1tag, IntTag<0>{}, IntTag<1>{});
        else
            go2(atag, q0tag, q1tag, IntTag<0>{}, IntTag<0>{});
    };
    auto go_q = [&](auto atag) {
        if (kq0 == 1 && kq1 == 8) go(atag, IntTag<1>{}, IntTag<8>{});
        else if (kq0 == 1 && kq1 == 4) go(atag, IntTag<1>{}, IntTag<4>{});
        else if (kq0 == 1 && kq1 == 2) go(atag, IntTag<1>{}, IntTag<2>{});
        else if (kq0 == 2 && kq1 == 8) go(atag, IntTag<2>{}, IntTag<8>{});
        else if (kq0 == 2 && kq1 == 4) go(atag, IntTag<2>{}, IntTag<4>{});
        else go(atag, IntTag<2>{}, IntTag<2>{});
    };
    GNNB_DISPATCH_ACT(act, go_q)
    return rc;
}

// GNNModel.output_activation (models.py:500-502, 572-573): softmax / log_softmax over each graph's output row.
// OUT is a handful of values (1..19 at the BASELINE configs): one lane per graph, three passes over the row.
__global__ __launch_bounds__(WG) void k_output_activation(float *__restrict__ out, int B, int n, int kind)
{
    const int g = blockIdx.x * WG + threadIdx.x;
    if (g >= B)
        return;
    float *o = out + (size_t)g * n;
    float mx = o[0];
    for (int i = 1; i < n; i++)
        mx = fmaxf(mx, o[i]);
    float sum = 0.0f;
    for (int i = 0; i < n; i++)
        sum += expf(o[i] - mx);
    for (int i = 0; i < n; i++)
        o[i] = kind == GNNB_OUT_SOFTMAX ? expf(o[i] - mx) / sum : (o[i] - mx) - logf(sum);
}

hipError_t launch_output_activation(float *out, int num_graphs, int n, int kind, hipStream_t s)
{
    if (num_graphs <= 0 || kind == GNNB_OUT_NONE)
        return hipSuccess;
    hipLaunchKernelGGL(k_output_activation, dim3((num_graphs + WG - 1) / WG), dim3(WG), 0, s, out, num_graphs, n, kind);
    return hipGetLastError();
}

// ap_fixed<W, I, AP_TRN, AP_WRAP> grid (reference code_gen.py:39-52, model.h.jinja:41-45): truncate towards minus
// infinity to a multiple of 2^-(W-I), wrap into [-2^(I-1), 2^(I-1)).  `inv_step` = 2^(W-I), `span` = 2^I.
__global__ __launch_bounds__(WG) void k_quantize(const float *__restrict__ src, float *__restrict__ dst, size_t n,
                                                 float inv_step, float step, float half_span, float span)
{
    for (size_t i = blockIdx.x * (size_t)WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
        float v = floorf(src[i] * inv_step) * step;
        v = v - span * floorf((v + half_span) / span); // two's-complement wrap
        dst[i] = v;
    }
}

hipError_t launch_quantize(const float *src, float *dst, size_t n, int W, int I, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const float inv_step = ldexpf(1.0f, W - I), step = ldexpf(1.0f, -(W - I));
    const float span = ldexpf(1.0f, I), half = ldexpf(1.0f, I - 1);
    const unsigned grid = (unsigned)std::min<size_t>((n + WG - 1) / WG, 4096);
    hipLaunchKernelGGL(k_quantize, dim3(grid), dim3(WG), 0, s, src, dst, n, inv_step, step, half, span);
    return hipGetLastError();
}

#ifdef GNNB_PROBE
extern "C" int gnnb_probe_read(unsigned long long *host, int count)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe), sizeof(unsigned long long) * count);
}
#endif

} // namespace gnnb
