// k_stack.hip -- whole GCN / GIN conv stack + pooling in one persistent kernel (k_gcn2_fused)
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include "gnnb_stack.h"

namespace gnnb {

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
// One MFMA phase of the fused stack: v[k][r] = act(A . Wslice^T + bias) for the wave's 16 columns and
// the rows (rg + k nrg) * 16 + lg * 4 + r of its units k < NU (NU wave-uniform).  The units'
// accumulators are interleaved so that dependent MFMAs are >= 2 issues apart (NU == 1: the k range
// is split over two accumulators instead).
template <int ACT, int KQ, int NU, bool SWZ, bool PIN = false>
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
        if (PIN) // (no fragment load is hoisted further than one k block ahead: the register budget of the deep / GIN variants)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int k = 0; k < NU; k++) {
                const float av = t == 0 ? a4[k].x : (t == 1 ? a4[k].y : (t == 2 ? a4[k].z : a4[k].w));
                const int ai = NU == 1 ? (t & 1) : k;
                acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[q * 4 + t], acc[ai], 0, 0, 0);
            }
        if (PIN)
            __builtin_amdgcn_sched_barrier(0);
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

// The same product with the units taken in groups of at most two (SPLIT): the 128-wide products of the deep / GIN
// variants hold a 32-register weight slice, the accumulators kept across a barrier AND the next slice's loads at once --
// four units in one go (32 fragment registers in flight) overflowed the 128-register budget into scratch (40-67 VGPRs
// spilled per instantiation, 23 MB of scratch writes per launch at BASELINE config 3).  Two units keep the dependent
// MFMAs of one accumulator 64 cycles apart (latency 40): the matrix pipe stays fed.
template <int ACT, int KQ, int NU, bool SPLIT>
__device__ __forceinline__ void g2_mma_s(const float *__restrict__ Asrc, int lda, const float (&wr)[KQ * 4], float bias, int rg,
                                         int nrg, int li, int lg, float (&v)[NU][4])
{
    if constexpr (SPLIT && NU > 2) {
        float va[2][4], vb[NU - 2][4];
        g2_mma<ACT, KQ, 2, false, true>(Asrc, lda, 1, wr, bias, rg, nrg, li, lg, va);
        __builtin_amdgcn_sched_barrier(0); // (the second group's fragment loads stay behind the first group's products)
        g2_mma<ACT, KQ, NU - 2, false, true>(Asrc, lda, 1, wr, bias, rg + 2 * nrg, nrg, li, lg, vb);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[0][r] = va[0][r];
            v[1][r] = va[1][r];
#pragma unroll
            for (int k = 2; k < NU; k++)
                v[k][r] = vb[k - 2][r];
        }
    } else {
        g2_mma<ACT, KQ, NU, false, SPLIT>(Asrc, lda, 1, wr, bias, rg, nrg, li, lg, v);
    }
}

// ---- "f16x3" (H3; opt-in, REDUCED precision: gnnb_set_option("math", 3)) for the deep / GIN variants: the A operand of every
// 128-wide product lives in LDS as hi + mid fp16 pieces, two planes inside the SAME padded fp32 row ([hi: h0 x 2 B][mid: h0 x 2 B]
// [pad]; the producer -- P1, or the product before -- writes them), the weight slice as hi + mid pieces in the same 4 KQ
// registers; three v_mfma_f32_16x16x32_f16 per 32-wide k block (mid.hi... hi.mid, hi.hi) instead of eight fp32 MFMAs of twice
// the passes.  Chunks of rows 4..11 (mod 16) are stored with the lowest bit of their index flipped (k_stack_zf.hip: conflict-
// free fragment reads); ONE fragment buffer (block q + 1 requested behind block q's MFMAs: the register budget is 128).
__device__ __forceinline__ int g2_h3_key(int row) { return ((row & 15) + 4) >> 3 & 1; }
// byte offset of element (row, column c) inside its plane (add row * row bytes, + 2 h0 for the mid plane)
__device__ __forceinline__ int g2_h3_off(int row, int c) { return 16 * ((c >> 3) ^ g2_h3_key(row)) + 2 * (c & 7); }
__device__ __forceinline__ void g2_h3_store(char *buf, int rowb, int midoff, int row, int c, float x)
{
    const _Float16 h = (_Float16)x, m = (_Float16)(x - (float)h);
    char *p = buf + row * rowb + g2_h3_off(row, c);
    *reinterpret_cast<_Float16 *>(p) = h;
    *reinterpret_cast<_Float16 *>(p + midoff) = m;
}
#ifndef GNNB_H3_PROBE // (development: 0 = no probe, 1 / 2 / 3 / 4 / 5 = forms measured against the register budget, see below)
#define GNNB_H3_PROBE 6
#endif
template <int ACT, int KQ32, int NU>
__device__ __forceinline__ void g2_mma_h3(const char *__restrict__ planes, int rowb, int midoff, const float (&wr)[KQ32 * 8], float bias,
                                          int rg, int nrg, int li, int lg, float (&v)[NU][4], int *sflag, int rows)
{
    constexpr int NA = NU == 1 ? 2 : NU;
    f32x4 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; a++)
        acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char *ap[NU];
    {
        int lane_off = li * rowb + ((lg ^ g2_h3_key(li)) << 4);
        asm volatile("" : "+v"(lane_off));
#pragma unroll
        for (int k = 0; k < NU; k++)
            ap[k] = planes + (rg + k * nrg) * 16 * rowb + lane_off;
    }
    u32x4 ah[NU], am[NU];
#pragma unroll
    for (int k = 0; k < NU; k++) {
        ah[k] = *reinterpret_cast<const u32x4 *>(ap[k]);
        am[k] = *reinterpret_cast<const u32x4 *>(ap[k] + midoff);
    }
#pragma unroll
    for (int q = 0; q < KQ32; q++) {
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 wh = {__float_as_uint(wr[q * 8 + 0]), __float_as_uint(wr[q * 8 + 1]), __float_as_uint(wr[q * 8 + 2]), __float_as_uint(wr[q * 8 + 3])};
        const u32x4 wm = {__float_as_uint(wr[q * 8 + 4]), __float_as_uint(wr[q * 8 + 5]), __float_as_uint(wr[q * 8 + 6]), __float_as_uint(wr[q * 8 + 7])};
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[NU == 1 ? 1 : k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(am[k]), as_f16x8(wh), acc[NU == 1 ? 1 : k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(ah[k]), as_f16x8(wm), acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[NU == 1 ? 1 : k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(ah[k]), as_f16x8(wh), acc[NU == 1 ? 1 : k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < KQ32) {
#pragma unroll
            for (int k = 0; k < NU; k++) {
                am[k] = *reinterpret_cast<const u32x4 *>(ap[k] + midoff + 64 * (q + 1));
                ah[k] = *reinterpret_cast<const u32x4 *>(ap[k] + 64 * (q + 1));
            }
        }
    }
    // (the reduced modes' overflow contract, gnnb_device.h: what the product gave, BEFORE the activation -- ReLU turns a NaN into 0 --,
    // in the rows the stage holds; rows past its end are stale LDS.  Seen -> one LDS word, which the kernel's last instructions turn
    // into the workspace's flag: nothing of the probe lives in a register outside this epilogue -- carried through the stage as a
    // scalar mask beside the flag word's address it spilled SGPRs into vector lanes in the variants that sit at 128 registers)
#if GNNB_H3_PROBE == 0
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            v[k][r] = act_t<ACT>((NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r]) + bias);
#elif GNNB_H3_PROBE == 1
    RangeProbe rp;
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float pre = (NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r]) + bias;
            rp.see(pre, (rg + k * nrg) * 16 + lg * 4 + r < rows);
            v[k][r] = act_t<ACT>(pre);
        }
    if (rp.any() && li + lg == 0)
        *sflag = 1;
#elif GNNB_H3_PROBE == 2
    // one probe value per lane: t stays 0 while every accumulator of the stage's rows is finite (inf * 0 and nan * 0 are nan)
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float pre = (NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r]) + bias;
            t = __builtin_fmaf((rg + k * nrg) * 16 + lg * 4 + r < rows ? pre : 0.0f, 0.0f, t);
            v[k][r] = act_t<ACT>(pre);
        }
    if (t != t)
        *sflag = 1;
#elif GNNB_H3_PROBE == 4 // experiment: no row mask
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float pre = (NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r]) + bias;
            t = __builtin_fmaf(pre, 0.0f, t);
            v[k][r] = act_t<ACT>(pre);
        }
    if (t != t)
        *sflag = 1;
#elif GNNB_H3_PROBE == 5 // experiment: no row mask, probe folded into v (no branch)
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float pre = (NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r]) + bias;
            t = __builtin_fmaf(pre, 0.0f, t);
            v[k][r] = act_t<ACT>(pre);
        }
    asm volatile("" :: "v"(t));
#elif GNNB_H3_PROBE == 6
    // one probe value per lane over ALL sixteen rows of the units (t stays 0 while every accumulator is finite: inf * 0 and
    // nan * 0 are nan); only when that trips -- rare -- the rows are looked at one by one against the stage's end (rows past it
    // are stale LDS): the row masks, a scalar register pair each, exist inside that branch only
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            t = __builtin_fmaf(NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r], 0.0f, t);
    if (__ballot(t != t) != 0ull) {
        RangeProbe rp;
#pragma unroll
        for (int k = 0; k < NU; k++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                rp.see(NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r], (rg + k * nrg) * 16 + lg * 4 + r < rows);
        if (rp.any() && li + lg == 0)
            *sflag = 1;
    }
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            v[k][r] = act_t<ACT>((NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r]) + bias);
#elif GNNB_H3_PROBE == 3
    // as 2, the row mask formed arithmetically (no compare: every v_cmp result is a scalar register pair, and these variants spill those)
    float t = 0.0f;
    const int lim = rows - rg * 16 - lg * 4; // rows of the stage from this lane's first row of unit 0 on
#pragma unroll
    for (int k = 0; k < NU; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float pre = (NU == 1 ? acc[0][r] + acc[1][r] : acc[k][r]) + bias;
            const int m = (k * nrg * 16 + r - lim) >> 31; // all ones while the row is inside the stage
            t = __builtin_fmaf(__int_as_float(__float_as_int(pre) & m), 0.0f, t);
            v[k][r] = act_t<ACT>(pre);
        }
    if (t != t)
        *sflag = 1;
#endif
}
// (units in groups of at most two, as g2_mma_s)
template <int ACT, int KQ32, int NU>
__device__ __forceinline__ void g2_mma_h3_s(const char *__restrict__ planes, int rowb, int midoff, const float (&wr)[KQ32 * 8], float bias,
                                            int rg, int nrg, int li, int lg, float (&v)[NU][4], int *sflag, int rows)
{
    if constexpr (NU > 2) {
        float va[2][4], vb[NU - 2][4];
        g2_mma_h3<ACT, KQ32, 2>(planes, rowb, midoff, wr, bias, rg, nrg, li, lg, va, sflag, rows);
        __builtin_amdgcn_sched_barrier(0);
        g2_mma_h3<ACT, KQ32, NU - 2>(planes, rowb, midoff, wr, bias, rg + 2 * nrg, nrg, li, lg, vb, sflag, rows);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[0][r] = va[0][r];
            v[1][r] = va[1][r];
#pragma unroll
            for (int k = 2; k < NU; k++)
                v[k][r] = vb[k - 2][r];
        }
    } else {
        g2_mma_h3<ACT, KQ32, NU>(planes, rowb, midoff, wr, bias, rg, nrg, li, lg, v, sflag, rows);
    }
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
template <int ACT, int KQ0, int KQ1, int MATH, bool DEEP = false, bool GIN = false, bool H3 = false>
__global__ __launch_bounds__(G2_WG, 4) void k_gcn2_fused(
    const float *__restrict__ x, int f0, const int4 *__restrict__ node_rec,
    const int32_t *__restrict__ col, const float *__restrict__ dinv,
    const int32_t *__restrict__ tile_first, const int32_t *__restrict__ tile_graph,
    const int32_t *__restrict__ node_ptr, int num_tiles, int num_graphs, int N, const float *__restrict__ W0,
    const float *__restrict__ b0, int h0, const float *__restrict__ W1, const float *__restrict__ b1,
    int h1, int p0, int p1, int p2, int np, float *__restrict__ pooled, int nl,
    const float *__restrict__ Wmid, const float *__restrict__ bmid, long mid_stride, long bmid_stride, int skip,
    float gin_eps, const int32_t *__restrict__ stage_cut,
    int32_t *__restrict__ err, int32_t *__restrict__ err_host) // H3: the workspace's flag word (GNNB_FLAG_RANGE, gnnb_device.h RangeProbe)
{
    const int h1out = h1; // the width of the pooled rows (GIN: the model's out_dim <= hidden; the products run hidden-wide)
    if (GIN) {
        // (GIN stacks are hidden x hidden everywhere -- the wide matrices come zero-padded, gnnb_model_create --, so the row strides of H and
        // A1 are compile-time constants: the sixteen write-back addresses of a wide product are one register + immediates
        // instead of sixteen hoisted registers, which is what pushed these variants over the 128-register budget)
        h0 = 16 * KQ1;
        h1 = 16 * KQ1;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!H3 || ((DEEP || GIN) && MATH == 0 && KQ1 % 2 == 0), "f16x3: the deep / GIN variants, whole 32-wide k blocks");
    constexpr int G2_UNITS = g2_units(MATH), G2_CAP = 16 * G2_UNITS;
    constexpr bool G2_SPLIT = (DEEP || GIN) && KQ1 >= 4; // wide products in groups of two units (register budget: g2_mma_s)
    constexpr bool G2_W0_PER_STAGE = DEEP || GIN || (MATH && KQ0 == 2); // the narrow slice re-read per stage (see M0)
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
    // (f16x3: "a reduced product of this workgroup gave a non-finite value" -- the last word of the first SMALL buffer's slack:
    // the DMA writes 65 of its 68 graph-boundary words)
    [[maybe_unused]] int *SFLAG = reinterpret_cast<int *>(smem + rows_b + small_b - 4);
    if (H3 && tid == 0)
        *SFLAG = 0;

    // the workgroup's run of node tiles: whole stages of the batch's global greedy stage list when graph prep made the cut
    // table for this grid (every workgroup the same number of stages +- 1, none ragged: round 5), else equal tile counts
    int t0, t1;
    run_cuts(blockIdx.x, gridDim.x, (unsigned)num_tiles, t0, t1); // (32-bit: gnnb_device.h)
    if (stage_cut && stage_cut[gridDim.x + 1] == 1) { // (clamped: a flagged batch's table may hold anything; the run must stay in range and below the LDS window)
        t0 = min(max(stage_cut[blockIdx.x], 0), num_tiles);
        t1 = min(max(stage_cut[blockIdx.x + 1], t0), min(num_tiles, t0 + G2_TCAP - 1));
    }
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
        st.nb = stile[min(ta, t1) - t0]; // (past the run: its end -- the stage before reads its own row count from there)
        if (ta >= t1)
            return st;
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
        if (!G2_W0_PER_STAGE && n0c < h0) // (else: loaded per stage, in front of M0)
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
    float bias0 = (!G2_W0_PER_STAGE && n0c < h0 && b0) ? b0[n0c] : 0.0f;
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
    if (!G2_W0_PER_STAGE) {
#pragma unroll
        for (int q = 0; q < KQ0 * 4; q++)
            asm volatile("" : "+v"(w0r[q]));
    }
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
    if (!G2_W0_PER_STAGE)
        asm volatile("" : "+v"(bias0));
    asm volatile("" : "+v"(bias1));
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
        // (f16x3: the reduced products' probe masks rows past the stage's end -- in a branch that runs only once something non-finite
        // was seen.  `rows` is uniform but sits in a VECTOR register (it comes from LDS): kept alive through the products it cost the
        // variants at the 128-register budget a spill; the difference of two values that ARE alive there is formed inside the branch)
        [[maybe_unused]] const int rows_p = min(nxt.nb - nb, G2_CAP);
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
            if constexpr (H3) {
                // (f16x3: per 32-wide k block the lane's eight k values 32 q + 8 lg .. + 7 of weight row ncol)
#pragma unroll
                for (int q = 0; q < KQ1 / 2; q++) {
                    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                    if (ncol < nlim) {
                        v0 = *reinterpret_cast<const float4 *>(Wl + (size_t)ncol * h0 + 32 * q + 8 * lg);
                        v1 = *reinterpret_cast<const float4 *>(Wl + (size_t)ncol * h0 + 32 * q + 8 * lg + 4);
                    }
                    // (RAW here: the loads stay in flight behind P1 / the write-back as in the fp32 form; split_slice()
                    // turns them into pieces right in front of the product)
                    w1r[q * 8 + 0] = v0.x, w1r[q * 8 + 1] = v0.y, w1r[q * 8 + 2] = v0.z, w1r[q * 8 + 3] = v0.w;
                    w1r[q * 8 + 4] = v1.x, w1r[q * 8 + 5] = v1.y, w1r[q * 8 + 6] = v1.z, w1r[q * 8 + 7] = v1.w;
                }
                bias1 = (ncol < nlim && bl) ? bl[ncol] : 0.0f;
                return;
            }
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
        // (f16x3) the raw slice -> hi + mid fp16 pieces, in place: ~20 instructions per k block, five slices per stage
        auto split_slice = [&]() {
            if constexpr (H3) {
#pragma unroll
                for (int q = 0; q < KQ1 / 2; q++) {
                    u32x4 hh, mm;
                    split2x8_f16(make_float4(w1r[q * 8 + 0], w1r[q * 8 + 1], w1r[q * 8 + 2], w1r[q * 8 + 3]),
                                 make_float4(w1r[q * 8 + 4], w1r[q * 8 + 5], w1r[q * 8 + 6], w1r[q * 8 + 7]), hh, mm);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        w1r[q * 8 + i] = __uint_as_float(hh[i]);
                        w1r[q * 8 + 4 + i] = __uint_as_float(mm[i]);
                    }
                }
            }
        };
        if (GIN)
            load_slice(Wmid, bmid, n0c, h0); // layer 0's second linear: in flight behind M0

        // ---- M0: H = act(A0 . W0^T + b0)   (wave: column slice x row group)
        {
            if (G2_W0_PER_STAGE) {
                // (deep / GIN variants, and the bf16x6 form with two narrow k blocks: the narrow slice and its bias are re-read from L2 per stage -- 16 + 4 bytes per lane,
                // requested here and first used behind the fragment read -- instead of living in five registers through the
                // wide phases, where the budget is 128: together with the two-unit products this took the variants from
                // 40-67 spilled registers to none)
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
                bias0 = (n0c < h0 && b0) ? b0[n0c] : 0.0f;
            }
            const int nu = rg0 < units ? (units - rg0 + nrg0 - 1) / nrg0 : 0;
            auto m0 = [&](auto nutag) {
                constexpr int NU = decltype(nutag)::value;
                float v[NU][4];
                g2_mma<GIN ? (int)GNNB_ACT_RELU : ACT, KQ0, NU, false>(A0, LD0, 1, w0r, bias0, rg0, nrg0, li, lg, v);
                if (n0c < h0) {
#pragma unroll
                    for (int k = 0; k < NU; k++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if constexpr (H3 && GIN) // (the next product reads H: fp16 pieces)
                                g2_h3_store(reinterpret_cast<char *>(H), ldh * 4, 2 * h0, (rg0 + k * nrg0) * 16 + lg * 4 + r, n0c, v[k][r]);
                            else
                                H[((rg0 + k * nrg0) * 16 + lg * 4 + r) * ldh + n0c] = v[k][r];
                        }
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
                if constexpr (H3) {
                    // fp16 pieces: this lane's 4 values = half of a 16-B chunk of 8, in each of the two planes of row rA
                    const _Float16 h0_ = (_Float16)accA.v.x, h1_ = (_Float16)accA.v.y, h2_ = (_Float16)accA.v.z, h3_ = (_Float16)accA.v.w;
                    char *dstp = reinterpret_cast<char *>(A1) + rA * lda1 * 4 + g2_h3_off(rA, gl * 4);
                    *reinterpret_cast<uint2 *>(dstp) = make_uint2(pack_f16(h0_, h1_), pack_f16(h2_, h3_));
                    *reinterpret_cast<uint2 *>(dstp + 2 * h0) =
                        make_uint2(pack_f16((_Float16)(accA.v.x - (float)h0_), (_Float16)(accA.v.y - (float)h1_)),
                                   pack_f16((_Float16)(accA.v.z - (float)h2_), (_Float16)(accA.v.w - (float)h3_)));
                } else if (MATH) {
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
            split_slice();
            const int nu = rg0 < units ? (units - rg0 + nrg0 - 1) / nrg0 : 0;
            auto mm = [&](auto nutag) {
                constexpr int NU = decltype(nutag)::value;
                float v[NU][4];
                if constexpr (H3)
                    g2_mma_h3_s<GNNB_ACT_NONE, KQ1 / 2, NU>(reinterpret_cast<const char *>(A1), lda1 * 4, 2 * h0, w1r, bias1, rg0, nrg0, li, lg, v, SFLAG, rows_p);
                else
                    g2_mma_s<GNNB_ACT_NONE, KQ1, NU, G2_SPLIT>(A1, lda1, w1r, bias1, rg0, nrg0, li, lg, v);
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
        auto m_inplace = [&](float *buf, int ld, auto acttag, int next_wide, auto outtag) {
            constexpr int A = decltype(acttag)::value;
            constexpr bool OUT_PIECES = H3 && decltype(outtag)::value != 0; // (f16x3: the NEXT product reads `buf` again)
            split_slice();
            const int nu = rg0 < units ? (units - rg0 + nrg0 - 1) / nrg0 : 0;
            float v[G2_UNITS][4];
            auto comp = [&](auto nutag) {
                constexpr int NU = decltype(nutag)::value;
                float t[NU][4];
                if constexpr (H3)
                    g2_mma_h3_s<A, KQ1 / 2, NU>(reinterpret_cast<const char *>(buf), ld * 4, 2 * h0, w1r, bias1, rg0, nrg0, li, lg, t, SFLAG, rows_p);
                else
                    g2_mma_s<A, KQ1, NU, G2_SPLIT>(buf, ld, w1r, bias1, rg0, nrg0, li, lg, t);
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
                        for (int r = 0; r < 4; r++) {
                            if constexpr (OUT_PIECES)
                                g2_h3_store(reinterpret_cast<char *>(buf), ld * 4, 2 * h0, (rg0 + k * nrg0) * 16 + lg * 4 + r, n0c, v[k][r]);
                            else
                                buf[((rg0 + k * nrg0) * 16 + lg * 4 + r) * ld + n0c] = v[k][r];
                        }
                    }
            }
            g2_barrier();
        };
        if (GIN) {
            // layer 0's second linear, then per further layer: aggregate, first linear (ReLU, in place on A1), second
            // linear (-> H with skip + activation; the LAST one stays in the accumulators for the pooling below)
            m_inplace(H, ldh, IntTag<ACT>{}, 1, IntTag<0>{}); // (its own slice, index 0, was requested in front of M0)
            G2_PT(6);
            for (int l = 1; l < nl; l++) {
                // (layer l's first slice, index 2l - 1, is in flight since the previous in-place product)
                phase_p1();
                G2_PT(8);
                g2_barrier();
                m_inplace(A1, lda1, IntTag<GNNB_ACT_RELU>{}, 2 * l, IntTag<1>{});
                G2_PT(9);
                if (l + 1 < nl) {
                    m_mid();
                    load_slice(Wmid + (size_t)(2 * l + 1) * mid_stride, bmid + (size_t)(2 * l + 1) * bmid_stride, n0c, h0);
                    g2_barrier();
                    G2_PT(7);
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
            split_slice();
            auto m1 = [&](auto nutag) {
                constexpr int NU = decltype(nutag)::value;
                float v[NU][4];
                if constexpr (H3)
                    g2_mma_h3_s<ACT, KQ1 / 2, NU>(reinterpret_cast<const char *>(A1), lda1 * 4, 2 * h0, w1r, bias1, 0, 1, li, lg, v, SFLAG, rows_p);
                else if (MATH)
                    g2_mma_bf6<ACT, KB1, NU>(reinterpret_cast<const char *>(A1), plane_b, prow_b, wh, wm, wl, bias1, li, lg, v);
                else
                    g2_mma_s<ACT, KQ1, NU, G2_SPLIT>(A1, lda1, w1r, bias1, 0, 1, li, lg, v);
                const int ngr = cur.gb - cur.ga;
                // (one store instruction per graph and pool; none if the whole slice is past h1; the rare
                // paths below that read global memory only make the count conservative -- see the wait)
                stores_behind_dma = wv * 16 < h1out ? ngr * np : 0;
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
                    if (lg == 0 && n1c < h1out) {
#pragma unroll
                        for (int kk = 0; kk < 3; kk++) {
                            if (kk >= np)
                                break;
                            float rr = sum;
                            if (pools[kk] == GNNB_POOL_MEAN)
                                rr = n > 0 ? sum / (float)n : 0.0f;
                            else if (pools[kk] == GNNB_POOL_MAX)
                                rr = n > 0 ? mx : 0.0f;
                            pooled[((size_t)(cur.ga + gi) * np + kk) * h1out + n1c] = rr;
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
            for (int e = tv; e < (cur.gb - cur.ga) * np * h1out; e += 64)
                pooled[(size_t)cur.ga * np * h1out + e] = 0.0f;
        }
        G2_PT(10);
#ifdef GNNB_PROBE
        nst++;
#endif
        cur = nxt;
        b ^= 1;
    }
    if constexpr (H3) { // the reduced mode's overflow contract: GNNB_FLAG_RANGE into the workspace's flag word (gnnb_device.h)
        __syncthreads();
        if (tid == 0 && *SFLAG != 0 && err) {
            atomicOr(err, GNNB_FLAG_RANGE);
            if (err_host)
                *reinterpret_cast<volatile int32_t *>(err_host) = GNNB_FLAG_RANGE;
        }
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
int gcn2_fused_tile_window() { return G2_TCAP - 1; }
int gcn2_fused_grid(int num_tiles)
{
    const long long g = std::min<long long>(2LL * device_cu_count(), num_tiles);
    return (int)std::max<long long>(g, 1);
}

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
    // (the opt-in bf16x6 math mode exists for the plain two-layer GCN form only: deeper GCN stacks and GIN stacks run their
    // fp32 kernel in either mode -- the mode may never make a model slower by sending it down the layer-by-layer path)
    if (deep.nl < 2 || (deep.nl > 2 && (!deep.wmid || (((uintptr_t)deep.wmid) & 15) || (deep.mid_stride & 3))))
        return hipErrorNotSupported;
    // GIN stacks: fp32 mode, out <= hidden (the wide matrices come hidden x hidden, zero-padded: gnnb_model_create), biases present
    if (deep.gin && (h1 > h0 || !deep.wmid || !deep.bmid || (((uintptr_t)deep.wmid) & 15) || (deep.mid_stride & 3)))
        return hipErrorNotSupported;
    const int math = (launch_math() && deep.nl == 2 && !deep.gin) ? 1 : 0;
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
    auto go2 = [&](auto atag, auto q0tag, auto q1tag, auto mtag, auto dtag, auto h3tag) {
        constexpr int ACT = decltype(atag)::value, KQ0 = decltype(q0tag)::value, KQ1 = decltype(q1tag)::value;
        constexpr int MATH = decltype(mtag)::value;
        constexpr bool DEEP = decltype(dtag)::value != 0, GIN = decltype(dtag)::value == 2, H3 = decltype(h3tag)::value != 0;
        auto kern = k_gcn2_fused<ACT, KQ0, KQ1, MATH, DEEP, GIN, H3>;
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
        // (graph prep's stage cuts, when they were made for exactly this grid)
        const int32_t *cut = (t.stage_cut && t.stage_cut_n == (int)grid) ? t.stage_cut : nullptr;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G2_WG), lds, s, x, f0, t.node_rec, t.col, t.dinv,
                           t.tile_first, t.tile_graph, t.graph_ptr, t.num_tiles, t.num_graphs, t.num_nodes, w0, b0, h0, w1, b1, h1, p0, p1, p2,
                           num_pools, pooled, deep.nl, deep.wmid, deep.bmid, deep.mid_stride, deep.bmid_stride, deep.skip, deep.eps, cut, t.err, t.err_host_dev);
        rc = hipGetLastError();
    };
    auto go = [&](auto atag, auto q0tag, auto q1tag) {
        const bool h3 = launch_math() == 3; // (opt-in f16x3, REDUCED precision: the GIN and deep variants)
        if (deep.gin && h3)
            go2(atag, q0tag, q1tag, IntTag<0>{}, IntTag<2>{}, IntTag<1>{});
        else if (deep.gin)
            go2(atag, q0tag, q1tag, IntTag<0>{}, IntTag<2>{}, IntTag<0>{});
        else if (math)
            go2(atag, q0tag, q1tag, IntTag<1>{}, IntTag<0>{}, IntTag<0>{});
        else if (deep.nl > 2) {
            // (the deep variants have no register to spare -- GELU 125 of 128 in fp32, its f16x3 form spilled one; with the reduced
            // mode's overflow probe, round 6, the sigmoid / tanh forms at hidden 128 spill one too --: at hidden 128 only ReLU stacks
            // take f16x3, the others keep fp32 in the mode, which is never less accurate; narrower stacks have registers to spare)
            if constexpr (decltype(atag)::value == GNNB_ACT_RELU || (decltype(atag)::value != GNNB_ACT_GELU && decltype(q1tag)::value < 8)) {
                if (h3) {
                    go2(atag, q0tag, q1tag, IntTag<0>{}, IntTag<1>{}, IntTag<1>{});
                    return;
                }
            }
            go2(atag, q0tag, q1tag, IntTag<0>{}, IntTag<1>{}, IntTag<0>{});
        }
        else
            go2(atag, q0tag, q1tag, IntTag<0>{}, IntTag<0>{}, IntTag<0>{});
    };
    auto go_q = [&](auto atag) {
        if (kq0 == 1 && kq1 == 8) go(atag, IntTag<1>{}, IntTag<8>{});
        else if (kq0 == 1 && kq1 == 4) go(atag, IntTag<1>{}, IntTag<4>{});
        else if (kq0 == 1 && kq1 == 2) go(atag, IntTag<1>{}, IntTag<2>{});
        else if (kq0 == 2 && kq1 == 8) go(atag, IntTag<2>{}, IntTag<8>{});
        else if (kq0 == 2 && kq1 == 4) go(atag, IntTag<2>{}, IntTag<4>{});
        else go(atag, IntTag<2>{}, IntTag<2>{});
    };
#ifdef GNNB_DEV_DEEPH3 // development builds: the deep-GCN f16x3 variants that sit at the 128-register budget, nothing else
    if (act == GNNB_ACT_RELU)
        go2(IntTag<GNNB_ACT_RELU>{}, IntTag<1>{}, IntTag<8>{}, IntTag<0>{}, IntTag<1>{}, IntTag<1>{});
    else
        go2(IntTag<GNNB_ACT_SIGMOID>{}, IntTag<2>{}, IntTag<8>{}, IntTag<0>{}, IntTag<1>{}, IntTag<1>{});
#elif defined(GNNB_DEV_FAST) // development builds: only the BASELINE config 2 / 3 instantiations (seconds instead of minutes to compile)
    if (act == GNNB_ACT_RELU && kq0 == 1 && kq1 == 8 && !deep.gin && !math && deep.nl == 2)
        go2(IntTag<GNNB_ACT_RELU>{}, IntTag<1>{}, IntTag<8>{}, IntTag<0>{}, IntTag<0>{}, IntTag<0>{});
    else if (act == GNNB_ACT_RELU && kq0 == 1 && kq1 == 8 && deep.gin && launch_math() == 3)
        go2(IntTag<GNNB_ACT_RELU>{}, IntTag<1>{}, IntTag<8>{}, IntTag<0>{}, IntTag<2>{}, IntTag<1>{});
    else if (act == GNNB_ACT_RELU && kq0 == 1 && kq1 == 8 && deep.gin)
        go2(IntTag<GNNB_ACT_RELU>{}, IntTag<1>{}, IntTag<8>{}, IntTag<0>{}, IntTag<2>{}, IntTag<0>{});
#else
    GNNB_DISPATCH_ACT(act, go_q)
#endif
    return rc;
}

} // namespace gnnb
