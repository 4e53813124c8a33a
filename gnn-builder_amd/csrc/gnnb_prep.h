// gnnb_prep.h -- graph prep (COO -> CSR by destination, degree scalers, node tiles) as DEVICE FUNCTIONS: one wavefront per graph.
// Included by k_prep.hip (the stand-alone kernel k_graph_prep) and, round 6, by k_readout.hip: the readout kernel k_head_small can
// run the graph prep of the NEXT batch of its stream as extra workgroups (PrepParams, gnnb_internal.h) -- the separate launch
// cost the three-stream pipeline ~4.5 us of a 42-us step as a guest beside the stack kernels.
// Reference: compute_degree_tables + compute_neighbor_tables (gnn_builder_lib.h:1051-1124), edge-index table (:1126-1166).
#pragma once
#include "gnnb_device.h"

namespace gnnb {

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


// 1 / sqrt(1 + d) for the in-degrees the molecule path can meet (<= 64 edges), as correctly rounded fp32 divisions of
// correctly rounded fp32 square roots -- bit-identical to `1.0f / sqrtf(1.0f + d)` on the device and in the oracle (hex
// float literals; generated with numpy float32).  One load instead of the ~35 instructions of the IEEE sqrt + division.
static __device__ const float k_dinv_by_degree[65] = {
    0x1.0000000000000p+0f, 0x1.6a09e60000000p-1f, 0x1.279a740000000p-1f, 0x1.0000000000000p-1f, 0x1.c9f25c0000000p-2f,
    0x1.a20bd60000000p-2f, 0x1.8309200000000p-2f, 0x1.6a09e60000000p-2f, 0x1.5555560000000p-2f, 0x1.43d1360000000p-2f,
    0x1.34bf640000000p-2f, 0x1.279a740000000p-2f, 0x1.1c01aa0000000p-2f, 0x1.11acee0000000p-2f, 0x1.08654a0000000p-2f,
    0x1.0000000000000p-2f, 0x1.f0b6860000000p-3f, 0x1.e2b7e00000000p-3f, 0x1.d5d7ea0000000p-3f, 0x1.c9f25c0000000p-3f,
    0x1.bee9040000000p-3f, 0x1.b4a2940000000p-3f, 0x1.ab099a0000000p-3f, 0x1.a20bd60000000p-3f, 0x1.99999a0000000p-3f,
    0x1.91a5560000000p-3f, 0x1.8a23460000000p-3f, 0x1.8309200000000p-3f, 0x1.7c4dd60000000p-3f, 0x1.75e9740000000p-3f,
    0x1.6fd4e80000000p-3f, 0x1.6a09e60000000p-3f, 0x1.6482d40000000p-3f, 0x1.5f3aa80000000p-3f, 0x1.5a2cd80000000p-3f,
    0x1.5555560000000p-3f, 0x1.50b06a0000000p-3f, 0x1.4c3abe0000000p-3f, 0x1.47f1460000000p-3f, 0x1.43d1360000000p-3f,
    0x1.3fd8080000000p-3f, 0x1.3c03660000000p-3f, 0x1.38512c0000000p-3f, 0x1.34bf640000000p-3f, 0x1.314c3e0000000p-3f,
    0x1.2df60c0000000p-3f, 0x1.2abb440000000p-3f, 0x1.279a740000000p-3f, 0x1.24924a0000000p-3f, 0x1.21a1860000000p-3f,
    0x1.1ec7020000000p-3f, 0x1.1c01aa0000000p-3f, 0x1.19507e0000000p-3f, 0x1.16b2900000000p-3f, 0x1.1426fc0000000p-3f,
    0x1.11acee0000000p-3f, 0x1.0f43a40000000p-3f, 0x1.0cea620000000p-3f, 0x1.0aa07c0000000p-3f, 0x1.08654a0000000p-3f,
    0x1.0638320000000p-3f, 0x1.0418a40000000p-3f, 0x1.0206140000000p-3f, 0x1.0000000000000p-3f, 0x1.fc0bd80000000p-4f};

// What a wave that prepares SEVERAL graphs fetched for one of them up front (prep_graph_group below): the graph's five table
// entries and, when its clamped edge range holds <= 64 edges, lane l's edge.  nullptr = the graph's wave fetches for itself.
struct PrepFetched {
    int np_m1, np_0, np_p1; // node_ptr[g - 1] (0 for g == 0), node_ptr[g], node_ptr[g + 1]
    int ep_0, ep_p1;        // edge_ptr[g], edge_ptr[g + 1]
    bool has_edge;          // `edge` is coo[e0 + lane] for lane < ne (e0, ne: the clamped range, as prep_one_graph forms it)
    int2 edge;
};

// Molecule path (<= 64 nodes AND <= 64 edges: one lane per edge, one lane per node; QM9, ESOL, most of ogbg-molhiv).
// Written for INSTRUCTION COUNT: with batches in flight this kernel runs beside the conv-stack kernel of another batch and
// costs the pipeline what it issues.  Instead of one ballot per destination node (a loop of n iterations of
// ~12 vector + scalar instructions), the lanes are matched on the BITS of the destination index: ceil(log2 n) ballots give
// every edge lane the mask of the lanes with the same destination (rank among them = popcount below the lane: stable,
// lanes are in COO order) and, from the same ballots, every NODE lane the mask of its in-edges (degree = popcount).
__device__ __forceinline__ void prep_graph_small(
    const int2 *__restrict__ coo, int n0, int n1, int e0, int ne, int32_t *__restrict__ row_ptr, int32_t *__restrict__ col,
    int32_t *__restrict__ eid, int4 *__restrict__ node_rec, float *__restrict__ dinv, float *__restrict__ amp,
    float *__restrict__ att, float delta, int drop_self, int32_t *__restrict__ err, int32_t *__restrict__ err_host,
    int32_t *__restrict__ s_first, int lane, const PrepFetched *pre = nullptr)
{
    const int n = n1 - n0;
    // ---- this lane's edge; an edge that leaves its graph is an error and is dropped, a GCN self loop is dropped silently
    int src = n0, d = 0;
    bool keep = false, bad = false;
    if (lane < ne) {
        const int2 e = (pre && pre->has_edge) ? pre->edge : coo[e0 + lane];
        if (e.x < n0 || e.x >= n1 || e.y < n0 || e.y >= n1)
            bad = true;
        else if (!(drop_self && e.x == e.y)) {
            keep = true;
            src = e.x;
            d = e.y - n0;
        }
    }
    // ---- match on the bits of the destination: `same` = edge lanes with this lane's destination, `mine` = edge lanes
    // whose destination is THIS lane's node index
    const unsigned long long valid = __ballot(keep);
    unsigned long long same = valid, mine = valid;
    const int nbits = 32 - __builtin_clz(max(n - 1, 1)); // wave-uniform, <= 6
    for (int b = 0; b < nbits; b++) {
        const unsigned long long mb = __ballot(keep && ((d >> b) & 1));
        same &= ((d >> b) & 1) ? mb : ~mb;
        mine &= ((lane >> b) & 1) ? mb : ~mb;
    }
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0));
    const int deg = lane < n ? __popcll(mine) : 0;
    // ---- row starts: inclusive wave scan of the degrees over the node lanes
    const int incl = wave_scan_incl(deg);
    const int start = e0 + incl - deg;
    if (lane < n) {
        const int v = n0 + lane;
        row_ptr[v] = start;
        dinv[v] = k_dinv_by_degree[min(deg, 64)];
        if (delta > 0.0f) { // (delta <= 0: the model has no PNA layer, the scalers are not needed)
            const int dcl = deg < 1 ? 1 : deg; // gnn_builder_lib.h:1972-1982
            const float logd = logf((float)(dcl + 1));
            amp[v] = logd / delta;
            att[v] = delta / logd;
        }
        // default record: unused source slots alias the node itself
        *reinterpret_cast<int4 *>(s_first + lane * 4) = make_int4(v, v, v, v);
    }
    // (the exchange through s_first crosses lanes: wave_barrier alone is not a memory ordering at the IR level, so each
    // hand-over is a wavefront-scope release / acquire pair -- no instruction on the device, only a compiler ordering)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- scatter: col[start[dst] + rank] = src (one store per edge lane), first four sources -> the node's record
    const int st = __shfl(start, d, 64);
    if (keep) {
        col[st + rank] = src;
        eid[st + rank] = e0 + lane; // COO row of this CSR slot (gnn_builder_lib.h:1126-1166)
        if (rank < 4)
            s_first[d * 4 + rank] = src;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < n) {
        const int4 f = *reinterpret_cast<const int4 *>(s_first + lane * 4);
        node_rec[2 * (size_t)(n0 + lane)] = make_int4(start, deg, f.x, f.y);
        node_rec[2 * (size_t)(n0 + lane) + 1] = make_int4(f.z, f.w, 0, 0);
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

// One graph (g < B), or the batch's tail (g == B: the end entries of the tables, the rows no graph owns), by one wavefront.
// s_first_w: PREP_FAST_NODES x 4 ints of LDS owned by this wave.
// PREP_FAST_NODES: 256 (4 node chunks of 64 lanes) in general, 64 when the caller promises graphs of <= 64
// nodes -- 4 KB of LDS per workgroup instead of 16 KB, so that graph prep of the next batch fits on a CU
// beside two workgroups of the conv-stack kernel and the readout of the previous one.
template <int PREP_FAST_NODES>
__device__ __forceinline__ void prep_one_graph(const int2 *__restrict__ coo, const int32_t *__restrict__ node_ptr, const int32_t *__restrict__ edge_ptr, int B, int N, int E, int32_t *__restrict__ row_ptr, int32_t *__restrict__ col, int32_t *__restrict__ eid, int4 *__restrict__ node_rec, float *__restrict__ dinv, float *__restrict__ amp, float *__restrict__ att, float delta, int32_t *__restrict__ tile_first, int32_t *__restrict__ tile_edge, int32_t *__restrict__ tile_graph, int32_t *__restrict__ graph_ptr, int tile_rows, int num_tiles, int max_graph_nodes_hint, int promise_graphs, int large_n, int large_e, int drop_self, int32_t *__restrict__ err, int32_t *__restrict__ err_host, int4 *__restrict__ agg_cut, int cut_log2, int32_t *__restrict__ node_graph, int g, int lane, int32_t *s_first_w, const PrepFetched *pre = nullptr)
{
    // (pre: g < B only -- the batch's tail reads entries 0 and B)
    auto np_at = [&](int k) { return pre ? (k < g ? pre->np_m1 : (k == g ? pre->np_0 : pre->np_p1)) : node_ptr[k]; };
    auto ep_at = [&](int k) { return pre ? (k == g ? pre->ep_0 : pre->ep_p1) : edge_ptr[k]; };

    // ---- node tiles: tile_first[t] = min{ node_ptr[g'] : node_ptr[g'] >= t*tile_rows }
    {
        // clamped so that a malformed node_ptr (flagged below) cannot write out of range
        const int p = (g == B) ? N : min(max(np_at(g), 0), N);
        const int t_lo = (g == 0) ? 0 : max(min(max(np_at(g - 1), 0), N) / tile_rows + 1, 0);
        const int t_hi = (g == B) ? num_tiles : min(p / tile_rows, num_tiles);
        // edges are grouped by graph, so the CSR segment of graph g starts at edge_ptr[g]
        const int pe = (g == B) ? E : min(max(ep_at(g), 0), E);
        for (int t = t_lo + lane; t <= t_hi; t += 64) {
            tile_first[t] = p;
            tile_edge[t] = pe;
            tile_graph[t] = g;
        }
        if (lane == 0)
            graph_ptr[g] = p; // the clamped copy later kernels read
    }
    // The caller's large segment (gnnb_workspace_set_large_segment) names its first graph AND that graph's node / edge
    // offsets; the stack kernels run on rows [0, large_n) and the layer-wise half on the rest.  A triple that disagrees
    // with the ptr arrays of THIS batch (stale workspace state from the batch before) would leave the rows between the two
    // boundaries to neither half: flagged here, where both arrays are read anyway.
    if (large_n >= 0 && g == promise_graphs && lane == 0 && (np_at(g) != large_n || ep_at(g) != large_e))
        flag_batch(err, err_host, 16);
    // Containment of malformed batches: whatever node_ptr / edge_ptr hold, every row in [0, N) leaves this
    // kernel with a record that later kernels can follow without leaving the buffers -- start and start + deg
    // inside [0, E], sources inside [0, N).  A graph's ranges are CLAMPED instead of rejected (any row r < N lies
    // in some pair node_ptr[g] <= r < node_ptr[g+1] when node_ptr runs from 0 to N; rows before node_ptr[0] or
    // after node_ptr[B] are given empty records by the last wave), only edges inside the clamped node range are
    // accepted, and the results of a flagged batch are unspecified but in range.
    auto empty_rows = [&](int r0, int r1) {
        for (int v = r0 + lane; v < r1; v += 64) {
            row_ptr[v] = 0;
            if (node_graph)
                node_graph[v] = -1;
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
            if (agg_cut)
                agg_cut[1 << cut_log2] = make_int4(N, N, E, B); // (the end of the last range)
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
    int n0 = np_at(g), n1 = np_at(g + 1);
    int e0 = ep_at(g), e1 = ep_at(g + 1);
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
    if (node_graph) // (the pooling epilogue of the last layer's GEMM walks rows by graph id: launch_linear, PoolEpilogue)
        for (int v = n0 + lane; v < n1; v += 64)
            node_graph[v] = g;
    // Row-balanced ranges of the gather-aggregate workgroups (k_aggregate_ring): range b of 2^cut_log2 starts at row
    // floor(b N / 2^cut_log2), usually in the middle of a graph -- the wave of the graph that owns that row records the
    // graph's first row / CSR slot beside it (both neighbours stage the boundary graph, each reduces its own rows).  No
    // search: lane l tests candidate b_est - 1 + l around a float estimate, exactly.
    if (agg_cut) {
        int bb = (int)((float)n0 * (float)(1 << cut_log2) / (float)max(N, 1)) - 2; // (wave-uniform)
        do { // (one pass for any graph of less than ~60 ranges' worth of rows)
            const int b = bb + lane;
            if (b >= 0 && b < (1 << cut_log2)) {
                const int r = (int)(((long long)b * N) >> cut_log2);
                if (r >= n0 && r < n1)
                    agg_cut[b] = make_int4(r, n0, e0, g);
            }
            bb += 64;
        } while (bb < (1 << cut_log2) && (int)(((long long)max(bb, 0) * N) >> cut_log2) < n1);
    }
    if (max_graph_nodes_hint > 0 && n > max_graph_nodes_hint && g < promise_graphs && lane == 0)
        flag_batch(err, err_host, 8); // the caller's max_graph_nodes promise does not hold for this batch
    if (n > PREP_FAST_NODES || ne > PREP_FAST_EDGES) { // wave-uniform
        prep_graph_scan(coo, n0, n1, e0, e1, row_ptr, col, eid, node_rec, dinv, amp, att, delta, drop_self, err, err_host);
        return;
    }
    if (n <= 64 && ne <= 64) { // wave-uniform: the molecule path
        prep_graph_small(coo, n0, n1, e0, ne, row_ptr, col, eid, node_rec, dinv, amp, att, delta, drop_self, err, err_host,
                         s_first_w, lane, pre);
        GNNB_STAMP_END(3);
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
                int32_t *f = s_first_w + vl * 4;
                f[0] = v;
                f[1] = v;
                f[2] = v;
                f[3] = v;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
                    s_first_w[d * 4 + erank[c]] = es[c];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < NC; q++) {
        const int vl = q * 64 + lane;
        if (vl < n) {
            const int32_t *f = s_first_w + vl * 4;
            node_rec[2 * (size_t)(n0 + vl)] = make_int4(start[q], deg[q], f[0], f[1]);
            node_rec[2 * (size_t)(n0 + vl) + 1] = make_int4(f[2], f[3], 0, 0);
        }
    }
    GNNB_STAMP_END(3);
    if (bad)
        flag_batch(err, err_host, 4);
}

// a kernel's FIRST argument where it lies in the kernarg segment: a PrepParams that only some of the kernel's workgroups read (k_head_small)
__device__ __forceinline__ const PrepParams *kernarg_prep_params()
{
    return (const PrepParams *)__builtin_amdgcn_kernarg_segment_ptr(); // (constant address space -> generic: the loads stay scalar)
}

// the same from a PrepParams (what both kernels hold)
template <int PREP_FAST_NODES>
__device__ __forceinline__ void prep_one_graph(const PrepParams &p, int g, int lane, int32_t *s_first_w, const PrepFetched *pre = nullptr)
{
    prep_one_graph<PREP_FAST_NODES>(p.coo, p.node_ptr, p.edge_ptr, p.B, p.N, p.E, p.row_ptr, p.col, p.eid, p.node_rec, p.dinv, p.amp, p.att, p.delta, p.tile_first, p.tile_edge, p.tile_graph, p.graph_ptr, p.tile_rows, p.num_tiles, p.max_graph_nodes_hint, p.promise_graphs, p.large_n, p.large_e, p.drop_self, p.err, p.err_host, p.agg_cut, p.cut_log2, p.node_graph, g, lane, s_first_w, pre);
}

// G consecutive graphs g0 .. g0 + G - 1 (those <= B) by ONE wavefront, their fetches batched (round 6).  The prep of a graph is a
// chain of dependent round trips -- its table entries, then its edges, then a few hundred instructions and the stores -- that
// occupies a wave slot for ~6 us and the machine hardly at all.  Beside the stack kernel of another batch there is ONE wave slot
// per SIMD for every guest kernel (k_gcn2_zf: four waves of <= 104 registers per SIMD leave 96): at BASELINE config 2 (4096
// graphs) one graph per wave made the prep 1025 workgroups x ~6 us = 24 us of ALL 256 guest slots per step, the readout's 11 us
// on top, against a 37.5-us step -- the guests queued and the pipeline ran at 42 us per step.  Here every lane-parallel load
// serves the whole group (lane l: entry g0 - 1 + l of node_ptr, g0 + l of edge_ptr), then the G edge loads are in flight
// together: two round trips per G graphs instead of 2 G, a quarter of the workgroups at G = 4.
template <int PREP_FAST_NODES, int G>
__device__ __forceinline__ void prep_graph_group(const PrepParams &p, int g0, int lane, int32_t *s_first_w)
{
    static_assert(G >= 1 && G <= 16, "group size");
    if (g0 > p.B)
        return;
    // entries g0 - 1 .. g0 + G of node_ptr (lanes 0 .. G + 1) and g0 .. g0 + G of edge_ptr (lanes 0 .. G), clamped to [0, B]
    const int npv = p.node_ptr[min(max(g0 - 1 + lane, 0), p.B)];
    const int epv = p.edge_ptr[min(g0 + lane, p.B)];
    PrepFetched f[G];
#pragma unroll
    for (int k = 0; k < G; k++) {
        f[k].np_m1 = __builtin_amdgcn_readlane(npv, k);
        f[k].np_0 = __builtin_amdgcn_readlane(npv, k + 1);
        f[k].np_p1 = __builtin_amdgcn_readlane(npv, k + 2);
        f[k].ep_0 = __builtin_amdgcn_readlane(epv, k);
        f[k].ep_p1 = __builtin_amdgcn_readlane(epv, k + 1);
        // the clamped edge range, exactly as prep_one_graph forms it (a malformed range is flagged there)
        const int e0 = min(max(f[k].ep_0, 0), p.E);
        const int ne = max(min(max(f[k].ep_p1, 0), p.E) - e0, 0);
        f[k].has_edge = g0 + k < p.B && ne <= 64; // (wave-uniform)
        f[k].edge = make_int2(0, 0);
        if (f[k].has_edge && lane < ne)
            f[k].edge = p.coo[e0 + lane];
    }
#pragma unroll
    for (int k = 0; k < G; k++) {
        const int g = g0 + k;
        if (g < p.B)
            prep_one_graph<PREP_FAST_NODES>(p, g, lane, s_first_w, &f[k]);
        else if (g == p.B)
            prep_one_graph<PREP_FAST_NODES>(p, g, lane, s_first_w);
    }
}

} // namespace gnnb
