// k_plan.hip -- the conv-stack kernels' workgroup ranges as WHOLE stages of the batch's global greedy stage list (round 5)
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
//
// k_gcn2_fused (k_stack.hip) gives every workgroup a run of node tiles and packs it greedily into stages of <= 64 rows (whole
// graphs).  With equal tile counts per workgroup every run ends in a ragged last stage and the runs' stage counts differ: at
// BASELINE config 3 (104 k rows, 512 resident workgroups) 4.14 stages per workgroup on average, 74 workgroups with five -- and
// the kernel ends with those.  Packed over the WHOLE batch the same rows are 2004 stages = 3.91 per workgroup: cut at every
// (S / G)-th stage boundary, every workgroup runs three or four stages and none is ragged (238 -> 216 us per launch).
// The greedy chain is sequential by nature (a stage begins where the one before ends); here it is resolved by binary lifting:
//   level 0   next[t] = the tile a stage that begins at tile t ends at (binary search in tile_first; nt = the absorbing end)
//   level k   jump_k[t] = jump_{k-1}[jump_{k-1}[t]]   (2^k stages ahead), ceil(log2(nt + 1)) rounds, one barrier each
//   S         = stages of the chain from tile 0 (greedy descent over the levels)
//   cut[b]    = the (b S / G)-th stage boundary = a walk over the set bits of that rank;  cut[G + 1] = 1 when every run fits the
//               kernel's tile-table window (else the kernel keeps its equal tile counts)
// ONE workgroup, no LDS, 18 registers.  MEASURED (round 5, BASELINE config 3): the stack kernel alone 232 -> 215 us per launch
// (0.47 -> 0.51 of the fp32 MFMA peak), one prepared forward on one stream 241 -> 222 us -- but the three-stream PIPELINE 206 -> 236
// us per step with sixteen waves and 286 with four (which would fit a free wave slot beside the stack kernel: it is not
// placed there): the planner's 35-80 us sit between graph prep and the stack kernel of every batch, and in the pipeline
// consecutive stack kernels already fill each other's ragged ends (206 us per step against a 232-us kernel).  Hence OFF by
// default (`stage_cut`); it pays where forwards run one at a time with the topology prepared once.  Reference: none (the
// reference runs one graph per call).
#include "gnnb_device.h"

namespace gnnb {

static constexpr int PL_WG = 1024;

__global__ __launch_bounds__(PL_WG) void k_stage_cut(const int32_t *__restrict__ tile_first, int nt, int N, int cap, int G, int tcap,
                                                     int32_t *__restrict__ jump, int K, int32_t *__restrict__ cut)
{
    __builtin_amdgcn_s_setprio(GNNB_GUEST_PRIO); // (co-runs with another batch's conv-stack kernel: see k_graph_prep)
    __shared__ int sS, sOk;
    const int tid = threadIdx.x;
    const int L = nt + 1;
    auto tf = [&](int t) { return min(max(tile_first[t], 0), N); }; // (clamped: a flagged batch's table may hold anything)
    // ---- level 0: the greedy end of a stage that begins at t (k_gcn2_fused's `plan`): the last tb with tf[tb] - tf[t] <= cap, at
    // least t + 1.  tile_first[tb] >= 8 tb - (largest graph) keeps the answer within a few dozen tiles: the search window is 64
    for (int t = tid; t <= nt; t += PL_WG) {
        int nx = nt;
        if (t < nt) {
            const int nb = tf(t);
            int lo = t + 1, hi = min(nt, t + 63);
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (tf(mid) - nb <= cap)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            nx = lo;
        }
        jump[t] = nx;
    }
    if (tid == 0)
        sOk = 1;
    __syncthreads();
    // ---- binary lifting (every level in its own array: nothing is read that this kernel wrote in the same round)
    // (the levels beyond the chain's length stay unwritten: K is cut to the first level whose jump from tile 0 reaches the end)
    for (int k = 0; k + 1 < K; k++) {
        const int32_t *a = jump + (size_t)k * L;
        if (a[0] >= nt) { // (workgroup-uniform: every thread reads the same word) 2^k stages from tile 0 pass the end
            K = k + 1;
            break;
        }
        int32_t *b = jump + (size_t)(k + 1) * L;
        for (int t = tid; t <= nt; t += PL_WG)
            b[t] = a[min(max(a[t], 0), nt)];
        __syncthreads();
    }
    // ---- the chain's length: the most hops from tile 0 that stay in front of the end, + the hop that reaches it
    if (tid == 0) {
        int t = 0, h = 0;
        for (int k = K - 1; k >= 0; k--) {
            const int j = jump[(size_t)k * L + t];
            if (j < nt) {
                t = j;
                h += 1 << k;
            }
        }
        sS = nt > 0 ? h + 1 : 0;
    }
    __syncthreads();
    const int S = sS;
    // ---- the cuts: workgroup b begins at the (b S / G)-th stage boundary
    for (int b = tid; b <= G; b += PL_WG) {
        const long long r = ((long long)b * S) / G;
        int t = 0;
        for (int k = 0; k < K; k++)
            if ((r >> k) & 1)
                t = jump[(size_t)k * L + min(max(t, 0), nt)];
        cut[b] = b == G ? nt : min(t, nt);
    }
    __syncthreads();
    for (int b = tid; b < G; b += PL_WG)
        if (cut[b + 1] - cut[b] > tcap || cut[b + 1] < cut[b])
            sOk = 0; // (a run beyond the kernel's tile-table window: it keeps its equal tile counts)
    __syncthreads();
    if (tid == 0)
        cut[G + 1] = sOk;
}

int stage_cut_levels(int max_tiles)
{
    int k = 1;
    while ((1ll << k) <= (long long)max_tiles + 1)
        k++;
    return k + 1;
}

// cut [G + 2] for `num_tiles` tiles into G runs of whole stages of <= cap rows; scratch: stage_cut_levels(num_tiles) x (num_tiles + 1) ints
hipError_t launch_stage_cut(const int32_t *tile_first, int num_tiles, int num_nodes, int cap, int G, int tcap, int32_t *scratch,
                            int32_t *cut, hipStream_t s)
{
    if (num_tiles <= 0 || G <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_stage_cut, dim3(1), dim3(PL_WG), 0, s, tile_first, num_tiles, num_nodes, cap, G, tcap, scratch,
                       stage_cut_levels(num_tiles), cut);
    return hipGetLastError();
}

} // namespace gnnb
