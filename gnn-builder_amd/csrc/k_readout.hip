// k_readout.hip -- per-graph add / mean / max pooling + MLP head: k_global_pool, k_pool_mlp, k_head_small
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include <algorithm>
#include <cstring>

#include "gnnb_device.h"
#include "gnnb_head.h"
#include "gnnb_prep.h"

namespace gnnb {

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
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

// GUEST (round 6, gnnb_forward_prepared_prep_next): the graph prep of the stream's NEXT batch -- COO -> CSR, node records,
// normalisers, tile tables of ANOTHER workspace (gnnb_prep.h: one wavefront per graph, <= 64 nodes each by that workspace's
// promise) -- as EXTRA WORKGROUPS of this launch: blocks [0, prep_blocks) prepare four graphs each, the rest run the head.  The
// prep is a chain of dependent fetches (its graph's table entries, its edges, then a few hundred instructions and the
// stores: ~6 us from launch to the last store) that occupies almost nothing; as a launch of its own beside the stack kernels
// of the other streams' batches it cost the three-stream pipeline ~4.5 us of a 42-us step.  Here it starts with the first
// blocks of a kernel that runs ~11 us anyway.  (Tried first inside k_gcn2_zf, on the waves that idle through the last stage's
// aggregation phase: that window is 2 us, the chain -- even with its fetches hoisted into the kernel's prologue -- outlasts
// it by 4 us, and the stack kernel is the pipeline's critical resource: no gain)
// The parameter block is the kernel's FIRST argument and is never named: the prep blocks read it from the kernarg segment
// (offset 0), the head blocks never fetch it.
// REGISTER BUDGET (round 6): 72 (a launch bound of seven waves per SIMD; 67 used).  Beside k_gcn2_zf's four 96-register waves a
// SIMD has 128 registers left: this kernel's wave and a graph-prep wave (56) of the stream's next batch then run side by side
// instead of one after the other -- BASELINE config 2, three batches in flight: 40.4-41.6 -> 37.6-37.7 us per step.
#ifndef GNNB_HS_WAVES
#define GNNB_HS_WAVES 7
#endif
// PAIRS = false (option head_pairs = 0): four operand slices in flight, 82 registers -- the faster form when NOTHING shares the chip (one
// stream of forwards: 49.8 vs 51.9 us per forward at BASELINE config 2), the slower one in the pipeline.
template <int ACT, bool GUEST, bool PAIRS = true>
__global__ __launch_bounds__(HS_THREADS, (GUEST || !PAIRS) ? 5 : GNNB_HS_WAVES) /* (the guest form carries the prep's code) */ void k_head_small(PrepParams guest_kernarg, int prep_blocks, const float *__restrict__ pooled, int B,
                                                             HeadArgs head, float *__restrict__ out, int ldact)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __builtin_amdgcn_s_setprio(GNNB_GUEST_PRIO); // (co-runs with the next batch's conv-stack kernel: see k_graph_prep)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (GUEST) {
        static_assert(HS_THREADS / 64 == WG / 64, "the guest prep's blocks are k_graph_prep's");
        if ((int)blockIdx.x < prep_blocks) {
            const PrepParams gp = *kernarg_prep_params();
            prep_graph_group<64, 4>(gp, (blockIdx.x * (HS_THREADS / 64) + wave) * 4, lane, reinterpret_cast<int32_t *>(smem) + wave * 256);
            return;
        }
    }
    float *sact = reinterpret_cast<float *>(smem); // [2][16][ldact]: ldact = widest hidden layer + 4 (padded rows)
    // one workgroup = one group of four waves = one tile of 16 graphs (gnnb_head.h)
    head_small_run<ACT, PAIRS ? 2 : 4>(pooled, 0, B, head, out, ldact, sact, lane, wave, (int)blockIdx.x - (GUEST ? prep_blocks : 0), 0, 1,
                                       [] { __syncthreads(); });
}

// hipErrorNotSupported when the head's shape does not suit the small form (caller takes k_pool_mlp)
static hipError_t launch_head_small(int num_graphs, const HeadArgs &head, int act, float *out, hipStream_t s,
                                    const float *prepooled)
{
    if (!prepooled || (((uintptr_t)prepooled) & 15))
        return hipErrorNotSupported;
    const int ldact = head_small_ldact(head);
    if (ldact <= 0)
        return hipErrorNotSupported;
    const int grid = (num_graphs + 15) / 16;
    // a guest graph prep on offer (gnnb_forward_prepared_prep_next)?
    GuestPrep *const gslot = guest_prep_slot();
    const bool guest = gslot && gslot->params && !gslot->taken && gslot->params->max_graph_nodes_hint > 0 && gslot->params->max_graph_nodes_hint <= 64;
    PrepParams gp;
    memset(&gp, 0, sizeof(gp));
    int prep_blocks = 0;
    if (guest) {
        gp = *gslot->params;
        prep_blocks = ((gp.B + 1 + 3) / 4 + (HS_THREADS / 64) - 1) / (HS_THREADS / 64);
    }
    const size_t lds = std::max(head_small_lds_bytes(ldact), guest ? (size_t)(HS_THREADS / 64) * 1024 : (size_t)0);
    auto go = [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
        if (guest)
            hipLaunchKernelGGL((k_head_small<ACT, true>), dim3(grid + prep_blocks), dim3(HS_THREADS), lds, s, gp, prep_blocks, prepooled, num_graphs, head, out, ldact);
        else if (options().head_pairs)
            hipLaunchKernelGGL((k_head_small<ACT, false>), dim3(grid), dim3(HS_THREADS), lds, s, gp, 0, prepooled, num_graphs, head, out, ldact);
        else
            hipLaunchKernelGGL((k_head_small<ACT, false, false>), dim3(grid), dim3(HS_THREADS), lds, s, gp, 0, prepooled, num_graphs, head, out, ldact);
    };
    GNNB_DISPATCH_ACT(act, go)
    const hipError_t rc = hipGetLastError();
    if (guest && rc == hipSuccess)
        gslot->taken = true;
    return rc;
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



} // namespace gnnb
