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
// ONE workgroup.  MEASURED (round 5, BASELINE config 3: 13 k tiles of 8 rows): the stack kernel alone 232 -> 215 us per launch
// (0.47 -> 0.51 of the fp32 MFMA peak), one prepared forward on one stream 240 -> 222 us -- but the three-stream PIPELINE only
// 202.0 -> 197.7 us per step with the cuts prepared for free (consecutive stack kernels already fill each other's ragged ends),
// and 203.5 -> 213 us with this launch between graph prep and the stack kernel (47 us alone in the LDS form below, 24-45 us with
// the tables in global memory; the first LDS form took 127 us alone -- found in the pipeline's kernel trace, not where it was
// written).  Hence OFF by default (`stage_cut`); it pays where forwards run one at a time.
// Reference: none (the reference runs one graph per call).
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

// The same plan with every table in LDS (<= 16383 tiles; 6 bytes per tile slot: 4 of tile starts -- the 16-bit rank table takes
// their place behind level 0 -- and 2 of one level, doubled in place through registers).  No level is kept: powers of one
// function commute, so "the tile 2^k stages behind a chain tile is a chain tile of rank + 2^k" can be applied level by level, LOW
// to high -- every round doubles the marked prefix of the chain (a tile marked in mid-round only marks further chain tiles with
// their true ranks: the races are benign) -- and when the levels are exhausted every chain tile knows its rank: S = the end's
// rank, and the chain tile of rank r is the start of every workgroup b with floor(b S / G) = r.
static constexpr int PLDS_MAX_TILES = 16383;
// TPT = tiles per thread (a compile-time bound keeps every per-thread table in registers and its LDS reads independent)
template <int TPT>
__global__ __launch_bounds__(PL_WG) void k_stage_cut_lds(const int32_t *__restrict__ tile_first, int nt, int N, int cap, int G, int tcap,
                                                         int32_t *__restrict__ cut)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int sOk, sMax;
    const int tid = threadIdx.x;
    int32_t *stf = reinterpret_cast<int32_t *>(smem);                                  // [nt + 1] tile starts (level 0 only)
    uint16_t *a = reinterpret_cast<uint16_t *>(smem + (size_t)TPT * PL_WG * 4);        // [nt + 1] level k: the tile 2^k stages behind t
    uint16_t *R = reinterpret_cast<uint16_t *>(smem);                                  // [nt + 1] rank of a chain tile, 0xffff = not (yet) known
    {
        int v[TPT];
#pragma unroll
        for (int c = 0; c < TPT; c++) {
            const int t = tid + c * PL_WG;
            v[c] = t <= nt ? tile_first[t] : 0;
        }
#pragma unroll
        for (int c = 0; c < TPT; c++) {
            const int t = tid + c * PL_WG;
            if (t <= nt)
                stf[t] = min(max(v[c], 0), N);
        }
    }
    if (tid == 0) {
        sOk = 1;
        sMax = 0;
    }
    __syncthreads();
    {
        // level 0: the greedy end of a stage that begins at t (k_gcn2_fused's `plan`): TPT binary searches side by side
        int lo[TPT], hi[TPT], nb[TPT];
#pragma unroll
        for (int c = 0; c < TPT; c++) {
            const int t = min(tid + c * PL_WG, nt);
            nb[c] = stf[t];
            lo[c] = min(t + 1, nt);
            hi[c] = min(nt, t + 63);
        }
        for (int step = 0; step < 6; step++) {
#pragma unroll
            for (int c = 0; c < TPT; c++) {
                const int mid = (lo[c] + hi[c] + 1) >> 1;
                const bool fits = stf[mid] - nb[c] <= cap;
                lo[c] = lo[c] < hi[c] && fits ? mid : lo[c];
                hi[c] = lo[c] < hi[c] && !fits ? mid - 1 : hi[c];
            }
        }
        __syncthreads(); // (the tile starts are dead: the ranks take their place)
#pragma unroll
        for (int c = 0; c < TPT; c++) {
            const int t = tid + c * PL_WG;
            if (t <= nt) {
                a[t] = (uint16_t)lo[c];
                R[t] = t == 0 ? 0 : 0xffff;
            }
        }
    }
    __syncthreads();
    for (int k = 0; k < 16; k++) {
        // ranks: a chain tile of rank r < 2^k hands rank r + 2^k to the tile 2^k stages behind it; and the next level's
        // values are READ (a is only read in this phase) ...
        int nx[TPT];
#pragma unroll
        for (int c = 0; c < TPT; c++) {
            const int t = min(tid + c * PL_WG, nt);
            const int j = a[t];
            const int r = R[t];
            nx[c] = a[j];
            const int rj = r + (1 << k);
            if (r != 0xffff && j < nt && t < nt && rj < 0xffff) // (the end is the jump's saturation value: it takes no rank from here)
                R[j] = (uint16_t)rj;                           // (several writers, one value: every path to a chain tile has the same length)
        }
        const bool done = a[0] >= nt; // (workgroup-uniform) 2^k stages from tile 0 pass the end: every chain tile is ranked after this round
        __syncthreads();
        if (done)
            break;
        // ... and written IN PLACE behind the barrier (a second level buffer would be another 2 (nt + 1) bytes)
#pragma unroll
        for (int c = 0; c < TPT; c++) {
            const int t = tid + c * PL_WG;
            if (t <= nt)
                a[t] = (uint16_t)nx[c];
        }
        __syncthreads();
    }
    // the number of stages = the last chain tile's rank + 1
    {
        int m = 0;
#pragma unroll
        for (int c = 0; c < TPT; c++) {
            const int t = tid + c * PL_WG;
            const int r = t < nt ? (int)R[t] : 0xffff;
            if (r != 0xffff)
                m = max(m, r + 1);
        }
        for (int o = 32; o; o >>= 1)
            m = max(m, __shfl_xor(m, o));
        if ((tid & 63) == 0 && m > 0)
            atomicMax(&sMax, m);
    }
    __syncthreads();
    const unsigned S = (unsigned)sMax;
    if (S == 0 || S >= 0xfff0u) { // (no chain, or one too long for the 16-bit ranks: no cuts)
        if (tid == 0)
            cut[G + 1] = 0;
        return;
    }
    // ---- the cuts: the chain tile of rank r starts every workgroup b with floor(b S / G) = r   (r G, b S < 2^26: 32-bit)
#pragma unroll
    for (int c = 0; c < TPT; c++) {
        const int t = tid + c * PL_WG;
        const unsigned r = t < nt ? (unsigned)R[t] : 0xffffu;
        if (r == 0xffffu)
            continue;
        for (unsigned bb = (r * (unsigned)G + S - 1) / S; bb < (unsigned)G && (bb * S) / (unsigned)G == r; bb++)
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
    if (num_tiles <= PLDS_MAX_TILES) { // every table in LDS: 6 bytes per tile slot
        hipError_t e = hipErrorNotSupported;
        auto go = [&](auto tag) {
            constexpr int TPT = decltype(tag)::value;
            const size_t lds = (size_t)TPT * PL_WG * 6;
            auto kern = k_stage_cut_lds<TPT>;
            e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
            if (e != hipSuccess)
                return;
            hipLaunchKernelGGL(kern, dim3(1), dim3(PL_WG), lds, s, tile_first, num_tiles, num_nodes, cap, G, tcap, cut);
            e = hipGetLastError();
        };
        if (num_tiles < 4 * PL_WG)
            go(IntTag<4>{});
        else if (num_tiles < 8 * PL_WG)
            go(IntTag<8>{});
        else
            go(IntTag<16>{});
        if (e == hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(k_stage_cut, dim3(1), dim3(PL_WG), 0, s, tile_first, num_tiles, num_nodes, cap, G, tcap, scratch,
                       stage_cut_levels(num_tiles), cut);
    return hipGetLastError();
}

} // namespace gnnb
