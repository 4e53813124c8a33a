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
// ONE workgroup.  MEASURED (round 5, BASELINE config 3): the stack kernel alone 232 -> 215 us per launch (0.47 -> 0.51 of the fp32
// MFMA peak), one prepared forward on one stream 241 -> 222 us -- but the three-stream PIPELINE 206 -> 236-246 us per step, whatever
// the planner's shape: tables in global memory and sixteen waves (~35 us alone), four waves (fit the free 32-register wave slot
// beside the stack kernel, and then crawl: ~230 us co-resident, 286 us per step), or every table in LDS (~10 us alone, but it
// needs a whole free CU, which the queued stack kernels of the other batches take first: 245 us per step).  A launch between
// graph prep and the stack kernel of every batch is on the pipeline's critical path, and there consecutive stack kernels
// already fill each other's ragged ends (206 us per step against a 232-us kernel).  Hence OFF by default (`stage_cut`); it pays
// where forwards run one at a time.  Reference: none (the reference runs one graph per call).
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

// The same plan with every table in LDS (<= 16383 tiles: 64 KB of tile starts -- a 32-KB rank table takes their place behind
// level 0 -- and one 32-KB level, doubled in place through registers): a round costs ~1 k cycles instead of two global round trips.  No level is kept: powers of one function
// commute, so "the tile 2^k stages behind a chain tile is a chain tile of rank + 2^k" can be applied level by level, LOW to high --
// every round doubles the marked prefix of the chain (a tile marked in mid-round only marks further chain tiles with their true
// ranks: the races are benign) -- and when the levels are exhausted every chain tile knows its rank: S = the end's rank, and the
// chain tile of rank r is the start of every workgroup b with floor(b S / G) = r.  ~10 us; it needs a free CU (128 KB of LDS),
// which the stack kernel of the batch before releases as its first workgroups end -- where the next stack kernel could not have
// started earlier either.
static constexpr int PLDS_MAX_TILES = 16383;
__global__ __launch_bounds__(PL_WG) void k_stage_cut_lds(const int32_t *__restrict__ tile_first, int nt, int N, int cap, int G, int tcap,
                                                         int32_t *__restrict__ cut)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int sOk, sMax;
    const int tid = threadIdx.x;
    int32_t *stf = reinterpret_cast<int32_t *>(smem);                      // [nt + 1] tile starts (level 0 only)
    uint16_t *A = reinterpret_cast<uint16_t *>(smem + 64 * 1024);          // [nt + 1] level k: the tile 2^k stages behind t
    uint16_t *R = reinterpret_cast<uint16_t *>(smem);                      // [nt + 1] rank of a chain tile, 0xffff = not (yet) known
    for (int t = tid; t <= nt; t += PL_WG)
        stf[t] = min(max(tile_first[t], 0), N);
    if (tid == 0) {
        sOk = 1;
        sMax = 0;
    }
    __syncthreads();
    for (int t = tid; t <= nt; t += PL_WG) {
        int nx = nt;
        if (t < nt) {
            const int nb = stf[t];
            int lo = t + 1, hi = min(nt, t + 63);
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (stf[mid] - nb <= cap)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            nx = lo;
        }
        A[t] = (uint16_t)nx;
    }
    __syncthreads(); // (the tile starts are dead: the ranks take their place)
    for (int t = tid; t <= nt; t += PL_WG)
        R[t] = t == 0 ? 0 : 0xffff;
    __syncthreads();
    uint16_t *a = A;
    for (int k = 0; k < 16; k++) {
        // ranks: a chain tile of rank r < 2^k hands rank r + 2^k to the tile 2^k stages behind it
        for (int t = tid; t < nt; t += PL_WG) {
            const int r = R[t];
            if (r != 0xffff) {
                const int j = a[t];
                const int rj = r + (1 << k);
                if (j < nt && rj < 0xffff) // (the end is the jump's saturation value: it takes no rank from here)
                    R[j] = (uint16_t)rj;   // (several writers, one value: every path to a chain tile has the same length)
            }
        }
        const bool done = a[0] >= nt; // (workgroup-uniform) 2^k stages from tile 0 pass the end: every chain tile is ranked after this round
        if (done)
            break;
        // the next level, doubled IN PLACE in two half-steps through registers (a second level buffer would be 32 KB more)
        __syncthreads();
        uint16_t nxt[(PLDS_MAX_TILES + 1 + PL_WG - 1) / PL_WG];
        int c = 0;
        for (int t = tid; t <= nt; t += PL_WG, c++)
            nxt[c] = a[a[t]];
        __syncthreads();
        c = 0;
        for (int t = tid; t <= nt; t += PL_WG, c++)
            a[t] = nxt[c];
        __syncthreads();
    }
    __syncthreads();
    // the number of stages = the last chain tile's rank + 1
    {
        int m = 0;
        for (int t = tid; t < nt; t += PL_WG) {
            const int r = R[t];
            if (r != 0xffff)
                m = max(m, r + 1);
        }
        if (m > 0)
            atomicMax(&sMax, m);
    }
    __syncthreads();
    const int S = sMax;
    if (S <= 0 || S >= 0xfff0) { // (no chain, or one too long for the 16-bit ranks: no cuts)
        if (tid == 0)
            cut[G + 1] = 0;
        return;
    }
    // ---- the cuts: the chain tile of rank r starts every workgroup b with floor(b S / G) = r
    for (int t = tid; t < nt; t += PL_WG) {
        const int r = R[t];
        if (r == 0xffff)
            continue;
        for (long long bb = ((long long)r * G + S - 1) / S; bb < G && (bb * S) / G == r; bb++)
            cut[bb] = t;
    }
    __syncthreads();
    if (tid == 0)
        cut[G] = nt;
    for (int bb = tid; bb < G; bb += PL_WG) {
        const int c0 = cut[bb], c1 = bb + 1 == G ? nt : cut[bb + 1];
        if (c1 - c0 > tcap || c1 < c0)
            sOk = 0; // (a run beyond the kernel's tile-table window: it keeps its equal tile counts)
    }
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
    if (num_tiles <= PLDS_MAX_TILES) { // every table in LDS
        const size_t lds = 96 * 1024;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_stage_cut_lds), lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_stage_cut_lds, dim3(1), dim3(PL_WG), lds, s, tile_first, num_tiles, num_nodes, cap, G, tcap, cut);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(k_stage_cut, dim3(1), dim3(PL_WG), 0, s, tile_first, num_tiles, num_nodes, cap, G, tcap, scratch,
                       stage_cut_levels(num_tiles), cut);
    return hipGetLastError();
}

} // namespace gnnb
