// gnnb_head.h -- the MLP head on pooled rows as a device function: k_head_small (k_readout.hip) runs it as a kernel of its
// own, k_gcn2_zf (k_stack_zf.hip) runs it at the end of every workgroup's life on the graphs that workgroup pooled (round 5:
// conv stack + pooling + head in ONE launch -- reference compute_gnn_head -> compute_global_graph_pooling ->
// compute_mlp_head inside one top, templates/model.cpp.jinja:737-765).
#pragma once
#include "gnnb_device.h"

namespace gnnb {

static constexpr int HS_MAXW = 128; // widest hidden layer this form takes

// Host side: does the head take this form?  (float4 operand fetches: widths % 4, 16-B aligned weights; hidden activations
// live in a 16 x ldact LDS tile.)  Returns ldact (floats per activation row) or 0.
static inline int head_small_ldact(const HeadArgs &head)
{
    if (head.nlin < 1 || head.nlin > 8)
        return 0;
    for (int l = 0; l < head.nlin; l++) {
        if ((head.dims[l] & 3) || (((uintptr_t)head.w[l]) & 15))
            return 0;
        if (l > 0 && head.dims[l] > HS_MAXW)
            return 0;
    }
    int maxw = 4;
    for (int l = 1; l < head.nlin; l++)
        maxw = std::max(maxw, (int)head.dims[l]);
    return ((maxw + 3) & ~3) + 4;
}
static inline size_t head_small_lds_bytes(int ldact) { return (size_t)2 * 16 * ldact * 4; } // per group of four waves

// One 16 x 16 output tile of one linear over the k range [k_lo, k_hi) (multiples of 16, or k): acc += A[16 graphs][k] . W[nn][k]^T,
// v_mfma_f32_16x16x4_f32, four accumulator chains over interleaved 16-wide k blocks, every operand fetch of a 64-wide k step in
// flight at once.  arow = this lane's A row (+ 4 lg), wrow = its W row (+ 4 lg); k = the row length (a float4 past it is zero).
__device__ __forceinline__ void head_tile_mma(const float *arow, const float *wrow, int k, int k_lo, int k_hi, f32x4 (&accs)[4])
{
    const int lg = (threadIdx.x & 63) >> 4;
    for (int kb = k_lo; kb < k_hi; kb += 64) {
        float4 a[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kk = kb + 16 * u; // (+ 4 lg inside the row pointers)
            const bool ok = kk < k_hi && kk + 4 * lg < k; // k % 4 == 0: a float4 is whole or absent
            const int kc = ok ? kk : 0;
            w[u] = *reinterpret_cast<const float4 *>(wrow + kc);
            a[u] = *reinterpret_cast<const float4 *>(arow + kc);
            if (!ok)
                a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
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
}

// The head at the END of a conv-stack workgroup's life (k_gcn2_zf), written for LATENCY: nothing else runs on the CU then, so
// every global round trip is paid in full.  The stand-alone form walks layer 0's K = np d (384 at BASELINE config 2) in six
// dependent steps of 64; here the workgroup's NGRP groups of four waves split that K between them -- each wave fetches its
// whole share at once: ONE round trip --, park their partial tiles in LDS and add them up in group order (deterministic).
// Later layers (K <= 128: one or two steps) run on group 0 as in the stand-alone form.  Tiles of 16 graphs one after the other
// (a workgroup pools 16 graphs at BASELINE config 2).  spart: LDS [NGRP][16][ldact], sact: LDS [2][16][ldact].
template <int ACT, int NGRP, typename Barrier>
__device__ __forceinline__ void head_tail_run(const float *__restrict__ pooled, int g_begin, int g_end, const HeadArgs &head,
                                              float *__restrict__ out, int ldact, float *spart, float *sact, int tid, Barrier bar)
{
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2, gw = wave & 3;
    const int ntile = (g_end - g_begin + 15) >> 4;
#pragma unroll 1
    for (int it = 0; it < ntile; it++) {
        const int g0 = g_begin + it * 16;
        const int grow = max(min(g0 + li, g_end - 1), 0); // (rows past the range re-read its last graph and are dropped at the store)
        // ---- layer 0, K split over the groups
        {
            const int k = head.dims[0], n = head.dims[1];
            const int kshare = (((k + 15) >> 4) + NGRP - 1) / NGRP * 16; // whole 16-wide blocks per group
            const int k_lo = min(grp * kshare, k), k_hi = min(k_lo + kshare, k);
            const float *__restrict__ W = head.w[0];
            for (int sl = gw; sl * 16 < n; sl += 4) {
                const int nn = sl * 16 + li;
                const int nnc = nn < n ? nn : n - 1;
                f32x4 accs[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                head_tile_mma(pooled + (size_t)grow * k + 4 * lg, W + (size_t)nnc * k + 4 * lg, k, k_lo, k_hi, accs);
                if (nn < n) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        spart[(grp * 16 + lg * 4 + r) * ldact + nn] = (accs[0][r] + accs[1][r]) + (accs[2][r] + accs[3][r]);
                }
            }
            bar();
            const float *__restrict__ bias = head.b[0];
            const bool last = head.nlin == 1;
            for (int e = tid; e < 16 * n; e += NGRP * 256) {
                const int gi = e / n, nn = e - gi * n;
                float v = spart[gi * ldact + nn];
#pragma unroll
                for (int g = 1; g < NGRP; g++)
                    v += spart[(g * 16 + gi) * ldact + nn];
                v += bias ? bias[nn] : 0.0f;
                if (last) {
                    if (g0 + gi < g_end)
                        out[(size_t)(g0 + gi) * n + nn] = v;
                } else {
                    sact[gi * ldact + nn] = act_t<ACT>(v);
                }
            }
            bar();
        }
        // ---- later layers on group 0 (activations in LDS)
        int cur = 0;
#pragma unroll 1
        for (int l = 1; l < head.nlin; l++) {
            const int k = head.dims[l], n = head.dims[l + 1];
            const bool last = (l == head.nlin - 1);
            const float *__restrict__ W = head.w[l];
            const float *__restrict__ bias = head.b[l];
            for (int sl = wave; sl * 16 < n; sl += NGRP * 4) { // (one 16-column slice per wave: n <= 128 fits sixteen or eight waves)
                const int nn = sl * 16 + li;
                const int nnc = nn < n ? nn : n - 1;
                f32x4 accs[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                head_tile_mma(sact + (cur * 16 + li) * ldact + 4 * lg, W + (size_t)nnc * k + 4 * lg, k, 0, k, accs);
                if (nn < n) {
                    const float bvv = bias ? bias[nn] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int gi = lg * 4 + r;
                        const float v = (accs[0][r] + accs[1][r]) + (accs[2][r] + accs[3][r]) + bvv;
                        if (last) {
                            if (g0 + gi < g_end)
                                out[(size_t)(g0 + gi) * n + nn] = v;
                        } else {
                            sact[((cur ^ 1) * 16 + gi) * ldact + nn] = act_t<ACT>(v);
                        }
                    }
                }
            }
            bar();
            cur ^= 1;
        }
    }
}

// One GROUP of four waves takes tiles of 16 graphs: tile index tile0 + it * tile_stride for it = 0 .. iters - 1, graphs
// g_begin + 16 tile ... (< g_end).  Every linear as 16 x 16 MFMA tiles (v_mfma_f32_16x16x4_f32): wave gw of the group takes the
// 16-column output slices gw, gw + 4, ...; A = the pooled rows straight from L2 (layer 0) or the activations in LDS, B = the
// weights straight from L2 (shared by every workgroup); four accumulator chains over interleaved 16-wide k blocks.
// `bar()` is the barrier that every wave of the WORKGROUP reaches once per layer and iteration -- groups without a tile
// still call it (iters is workgroup-uniform).  sact: this group's LDS tile [2][16][ldact].
// UW: k slices of a 64-wide block fetched and multiplied together -- 2 (default: the 72-register form, below) or 4 (82 registers, more
// loads in flight: ~2 us faster when the kernel has the chip to itself; option head_pairs = 0).  Same accumulators, same sums.
template <int ACT, int UW = 2, typename Barrier>
__device__ __forceinline__ void head_small_run(const float *__restrict__ pooled, int g_begin, int g_end, const HeadArgs &head,
                                               float *__restrict__ out, int ldact, float *sact, int lane, int gw, int tile0,
                                               int tile_stride, int iters, Barrier bar)
{
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        const int g0 = g_begin + (tile0 + it * tile_stride) * 16;
        const bool on = g0 < g_end;                          // (wave-uniform)
        const int grow = max(min(g0 + li, g_end - 1), 0);    // (rows past the range re-read its last graph and are dropped at the store)
        int cur = 0;
#pragma unroll 1
        for (int l = 0; l < head.nlin; l++) {
            const int k = head.dims[l], n = head.dims[l + 1];
            const bool last = (l == head.nlin - 1);
            const float *__restrict__ W = head.w[l];
            const float *__restrict__ bias = head.b[l];
            for (int sl = gw; on && sl * 16 < n; sl += 4) {
                const int nn = sl * 16 + li;
                const int nnc = nn < n ? nn : n - 1;
                const float *wrow = W + (size_t)nnc * k + 4 * lg;
                const float *arow_g = pooled + (size_t)grow * k + 4 * lg;          // layer 0: A straight from the pooled matrix
                const float *arow_l = sact + (cur * 16 + li) * ldact + 4 * lg;     // later layers: from LDS
                f32x4 accs[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    accs[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // The four k slices of a 64-wide block are fetched and multiplied as TWO PAIRS (round 6): half the operand registers of
                // the four-at-once form, the same four accumulators and the same sums in the same order (bit-identical results).
                // What it buys is the stand-alone kernel's register count, 82 -> 67: beside four 96-register waves of k_gcn2_zf a SIMD
                // has 128 registers left, and a 72-register readout wave + a 56-register graph-prep wave now fit TOGETHER.
                // (no operand double-buffering either: the register budget is what lets the stand-alone kernel share a SIMD with the
                // conv-stack kernel of another batch, and inside k_gcn2_zf it must stay below that kernel's own budget)
                static_assert(UW == 2 || UW == 4, "operand group width");
                for (int kb = 0; kb < k; kb += 64) {
#pragma unroll
                    for (int hp = 0; hp < 4 / UW; hp++) {
                        float4 a[UW], w[UW];
#pragma unroll
                        for (int u2 = 0; u2 < UW; u2++) {
                            const int kk = kb + 16 * (UW * hp + u2); // (+ 4 lg inside the row pointers)
                            const bool ok = kk + 4 * lg < k;         // k % 4 == 0: a float4 is whole or absent
                            const int kc = ok ? kk : 0;
                            w[u2] = *reinterpret_cast<const float4 *>(wrow + kc);
                            a[u2] = l == 0 ? *reinterpret_cast<const float4 *>(arow_g + kc)  // layer 0: A straight from the pooled matrix
                                           : *reinterpret_cast<const float4 *>(arow_l + kc); // later layers: from LDS
                            if (!ok)
                                a[u2] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
#pragma unroll
                        for (int u2 = 0; u2 < UW; u2++)
                            accs[UW * hp + u2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u2].x, w[u2].x, accs[UW * hp + u2], 0, 0, 0);
#pragma unroll
                        for (int u2 = 0; u2 < UW; u2++)
                            accs[UW * hp + u2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u2].y, w[u2].y, accs[UW * hp + u2], 0, 0, 0);
#pragma unroll
                        for (int u2 = 0; u2 < UW; u2++)
                            accs[UW * hp + u2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u2].z, w[u2].z, accs[UW * hp + u2], 0, 0, 0);
#pragma unroll
                        for (int u2 = 0; u2 < UW; u2++)
                            accs[UW * hp + u2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u2].w, w[u2].w, accs[UW * hp + u2], 0, 0, 0);
                    }
                }
                // C/D: col = lane&15 (output column nn), row = (lane>>4)*4 + r (graph inside the tile)
                if (nn < n) {
                    const float bvv = bias ? bias[nn] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int gi = lg * 4 + r;
                        const float v = (accs[0][r] + accs[1][r]) + (accs[2][r] + accs[3][r]) + bvv;
                        if (last) {
                            if (g0 + gi < g_end)
                                out[(size_t)(g0 + gi) * n + nn] = v;
                        } else {
                            sact[((cur ^ 1) * 16 + gi) * ldact + nn] = act_t<ACT>(v);
                        }
                    }
                }
            }
            bar();
            cur ^= 1;
        }
    }
}

} // namespace gnnb
