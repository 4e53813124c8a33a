// k_prep.hip -- graph prep: COO -> CSR-by-destination + degree scalers + node tiles (latency bound)
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include <cstring>

#include "gnnb_prep.h"

namespace gnnb {

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
template <int PREP_FAST_NODES, int GROUP>
__global__ __launch_bounds__(WG) void k_graph_prep(PrepParams p)
{
    __shared__ __attribute__((aligned(16))) int32_t s_first[WG / 64][PREP_FAST_NODES * 4]; // first four sources of every node (read / written 16 B at a time)
    // Highest wave priority: with batches in flight on several streams this kernel runs BESIDE the conv-stack kernel of
    // another batch (one wave slot per SIMD is left over there) and sits on its own stream's critical path -- 37-53 us
    // instead of 7 when it queues behind sixteen MFMA-issuing waves per CU.  It is a few hundred instructions per wave;
    // letting them issue first costs the big kernel nothing measurable (C2 step 55.5 -> 54.0 us).
    __builtin_amdgcn_s_setprio(GNNB_GUEST_PRIO);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (GROUP > 1)
        prep_graph_group<PREP_FAST_NODES, GROUP>(p, (blockIdx.x * (WG / 64) + wave) * GROUP, lane, s_first[wave]);
    else {
        const int g = blockIdx.x * (WG / 64) + wave;
        if (g > p.B)
            return;
        prep_one_graph<PREP_FAST_NODES>(p, g, lane, s_first[wave]);
    }
}

// the kernel arguments of one graph prep (also handed to the GCN stack kernel that runs the prep of its stream's next batch)
PrepParams make_prep_params(const int32_t *coo, const int32_t *node_ptr, const int32_t *edge_ptr, const BatchTables &t, float pna_delta,
                            int drop_self_loops)
{
    int cut_log2 = -1; // (row-balanced aggregate ranges: only for a power-of-two range count)
    if (t.agg_cut && t.agg_cut_n > 0 && (t.agg_cut_n & (t.agg_cut_n - 1)) == 0 && t.num_nodes > 0)
        for (cut_log2 = 0; (1 << cut_log2) < t.agg_cut_n; cut_log2++) {
        }
    PrepParams p;
    memset(&p, 0, sizeof(p)); // (padding too: the runtime compares parameter blocks bytewise)
    p.coo = (const int2 *)coo, p.node_ptr = node_ptr, p.edge_ptr = edge_ptr;
    p.B = t.num_graphs, p.N = t.num_nodes, p.E = t.num_edges;
    p.row_ptr = t.row_ptr, p.col = t.col, p.eid = t.eid, p.node_rec = t.node_rec, p.dinv = t.dinv, p.amp = t.amp, p.att = t.att;
    p.delta = pna_delta;
    p.tile_first = t.tile_first, p.tile_edge = t.tile_edge, p.tile_graph = t.tile_graph, p.graph_ptr = t.graph_ptr;
    p.tile_rows = t.tile_rows, p.num_tiles = t.num_tiles, p.max_graph_nodes_hint = t.max_graph_nodes_hint, p.promise_graphs = t.promise_graphs;
    p.large_n = t.large_n, p.large_e = t.large_e, p.drop_self = drop_self_loops, p.err = t.err, p.err_host = t.err_host_dev;
    p.agg_cut = cut_log2 >= 0 ? t.agg_cut : nullptr, p.cut_log2 = cut_log2, p.node_graph = t.node_graph;
    return p;
}

hipError_t launch_graph_prep(const PrepParams &p, hipStream_t s)
{
    // the workspace's flag word is zeroed when the workspace is created and again whenever it is read
    // (gnnb_workspace_check), so no per-batch memset node sits in front of this launch
    const int waves = p.B + 1;
    if (p.max_graph_nodes_hint > 0 && p.max_graph_nodes_hint <= 64) {
        // molecule-sized graphs: a wave prepares a GROUP of graphs with its fetches batched (prep_graph_group) once there are enough
        // graphs to fill the chip's guest wave slots several times over
        // (BASELINE config 2, 4096 graphs, three batches in flight: 41.2 instead of 42.3 us per step; ONE batch in flight, where the chip's
        // every wave slot is free, is slower with groups -- 58.9 instead of 55.7 us per forward: option prep_group = 1)
        const int group = (int)options().prep_group;
        if (group >= 4 && waves >= 2048) {
            const int grid = ((waves + 3) / 4 + (WG / 64) - 1) / (WG / 64);
            hipLaunchKernelGGL((k_graph_prep<64, 4>), dim3(grid), dim3(WG), 0, s, p);
        } else {
            const int grid = (waves + (WG / 64) - 1) / (WG / 64);
            hipLaunchKernelGGL((k_graph_prep<64, 1>), dim3(grid), dim3(WG), 0, s, p);
        }
    } else {
        const int grid = (waves + (WG / 64) - 1) / (WG / 64);
        hipLaunchKernelGGL((k_graph_prep<256, 1>), dim3(grid), dim3(WG), 0, s, p);
    }
    return hipGetLastError();
}

hipError_t launch_graph_prep(const int32_t *coo, const int32_t *node_ptr, const int32_t *edge_ptr,
                             BatchTables &t, float pna_delta, int drop_self_loops, hipStream_t s)
{
    return launch_graph_prep(make_prep_params(coo, node_ptr, edge_ptr, t, pna_delta, drop_self_loops), s);
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


} // namespace gnnb
