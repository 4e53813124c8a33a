// k_stack_zf.hip -- 2-layer GCN stack + pooling in one persistent kernel, last layer TRANSFORMED BEFORE it is aggregated
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include <cstring>

#include "gnnb_stack.h"
#include "gnnb_head.h"

// TWO translation units are made of this file (round 6): k_stack_zf.hip itself holds the kernels WITHOUT the MLP-head tail (the
// default forward: the head is a guest launch) and k_stack_zf_head.hip (#define ZF_TU_HEAD 1 + #include of this
// file) the ones WITH it (option zf_head).  Without the tail's code the fp32 kernels fit 96 registers, and five waves per SIMD's
// worth of budget leave a 128-register hole beside them -- room for TWO guest waves per SIMD (graph prep 56, readout 88) where
// 104 registers left one (see REGISTER BUDGET below).
#ifndef ZF_TU_HEAD
#define ZF_TU_HEAD 0
#endif

namespace gnnb {

#ifndef GNNB_ZF_KERNEL_DEFINED // (the probe build includes both translation units' sources into one)
#define GNNB_ZF_KERNEL_DEFINED
// =====================================================================================
// k_gcn2_zf: the BASELINE config 1 / 2 model family (two GCN layers, fp32), round 3
// =====================================================================================
// Reference dataflow being fused: compute_gnn_head with two gcn_conv layers (templates/model.cpp.jinja:151-359,
// gnn_builder_lib.h:1213-1387: aggregate, then `linear`, then the activation) + compute_global_graph_pooling
// (:413-449, global_*_pool lib:2709-2803).
//
// Difference to k_gcn2_fused (k_stack.hip), which keeps the reference's aggregate-then-transform order in both layers
// and needs a second [rows, h0] LDS matrix A1 for the aggregated hidden rows:
//   layer 1 here is   out = act( A^ . (H . W1^T) + b1 )   instead of   act( (A^ . H) . W1^T + b1 )
// -- the same mathematics up to fp32 summation order (both are a double sum over neighbours j and hidden units k).
// The product Z = H . W1^T is formed on the matrix cores straight from H, kept in the accumulators across one barrier
// and written back OVER H; the aggregation then runs on Z, one wave per GRAPH, each lane group walking its rows in
// order with bias / activation applied on the way and the add / mean / max pooling accumulated in registers.  What
// this buys on this chip (fp32 MFMA and VALU instructions share one issue port, so every VALU instruction is paid in
// matrix slots -- DESIGN 3.5):
//   * no A1 buffer: 34 KB less LDS per 64 rows, spent on BIGGER stages -- up to 96 rows (6 MFMA units) with two
//     workgroups per CU, so a workgroup's share of the BASELINE config 2 batch (~144 rows) is two stages instead of
//     three or four, and the fixed per-stage chain (DMA wait, four barriers, the two narrow phases) is paid less often;
//   * pooling without masks: the old form pooled the accumulator tiles of M1 (rows spread over registers and lane
//     groups, ~7 VALU per element and graph for the in / out-of-graph selects); here a graph's rows arrive one after
//     the other in one lane group: 2 VALU per element;
//   * stages are planned BALANCED (equal shares of the workgroup's rows, cut at graph boundaries) instead of greedily
//     filled: with a 96-row stage capacity against ~72 rows needed, every workgroup of the config 2 batch runs exactly
//     two stages and the kernel no longer ends with the one workgroup in ten that needed an extra stage;
//   * P0 of stage s+1 (the narrow aggregate of the raw features) runs in the same barrier interval as the wide
//     aggregate of stage s, on the waves that have no graph to reduce: four barriers per stage, not five.
//
// Per stage s (rows of whole graphs, <= 96):
//   top   issue DMA: raw x rows + node records of stage s+1 -> ROWS (single buffer: P0(s) is done), dinv + graph
//         boundaries of s+1 -> SMALL[(s+1)&1]                                                   (global_load_lds)
//   M0    H = act(A0 . W0^T + b0)                 MFMA 16x16x4, W0 slice in registers           A0 -> H
//   ---- barrier
//   M1    Z = H . W1^T                            MFMA, W1 slice (16 cols x K) in registers     H -> accumulators
//   wait  own DMA of stage s+1 landed (vmcnt(0): nothing younger is in flight)
//   ---- barrier   (everybody has read H; everybody's DMA is in)
//   ZW    Z -> H (in place)
//   ---- barrier
//   P1    per graph (one wave each): out_i = act(sum_j c_ij Z_j + c_ii Z_i + b1), pooled add / mean / max -> HBM
//   P0'   A0 = aggregate(x) of stage s+1 (eight lanes per row) + its per-row records REC[(s+1)&1]
//   ---- barrier
// HBM traffic = x + tables in, [B, np*h1] out (as k_gcn2_fused).  Bound: fp32 MFMA.
// Needs: GCN, exactly two layers, fp32 math mode, F0 <= 32, h0 in {32,64,128}, h1 <= 128 (h1 % 4 == 0), and the caller's
// promise max_graph_nodes <= 96 - (tile_rows - 1) (validated by graph prep).
// Two shapes (runtime option zf_shape; 2 = default = the first where it exists): 1 = ONE workgroup of 16 waves per CU,
// stages of up to 176 rows (11 MFMA units) -- a CU's share of the BASELINE config 2 batch (288 rows +- one graph) is always
// TWO stages (with a 160-row capacity one workgroup in a few hundred found no graph boundary inside the window that lets
// two stages hold its rows and ran a third: the kernel ends with its slowest workgroup), ~143 KB of LDS leave room for the
// readout / graph-prep kernels of the other batches in flight; 0 = two workgroups of 8 waves per CU, stages of up to 96
// rows (159 KB: nothing co-resides), the only shape for input widths of 17 .. 32.
#ifndef ZF_PRIO
#define ZF_PRIO 2
#endif
// build-time switches: measured alternatives of this kernel, kept compilable (same-box A/B on BASELINE config 2, wide shape:
// DESIGN 3.5a "round 4")
#ifndef ZF_K12         // input widths <= 12: three MFMA k steps per unit in M0 (k = lg + 4 t) instead of four: 38.9 -> 38.55 us
#define ZF_K12 1
#endif
#ifndef ZF_DMA_IN_M1   // the next stage's DMA issued behind the H barrier instead of at the stage top: +-0 (wide), -0.25 us (96-row shape)
#define ZF_DMA_IN_M1 1
#endif
#ifndef ZF_PIN         // scheduler barriers pin "request block q + 1's fragments, THEN issue block q's MFMAs" in zf_mma (left alone the
#define ZF_PIN 1       // compiler sinks every ds_read to just above its first use): 38.73 -> 38.55 us, 99 VGPRs
#endif
#ifndef ZF_SWZ         // H / Z rows unpadded, 16-B chunks XOR-swizzled by the row index: LDS bank conflicts 33 % -> 16 % of LDS cycles and the
#define ZF_SWZ 0       // kernel 1.7 us SLOWER (38.8 -> 40.5 us, 117 VGPRs): the conflicts sit in M1's fragment reads, where the
#endif                 // LDS array is < 25 % busy -- they cost nothing; the swizzle's address arithmetic does

// accumulate NU 16-row units (rows row0[k] + li) x the wave's 16-column slice over K = 16 KQ:
// acc[k] += Wslice . A[rows of unit k][:]^T -- the TRANSPOSED tile (weight fragment as the first MFMA operand), so that
// lane (li, lg) ends up with FOUR CONSECUTIVE columns 16 s + 4 lg .. + 3 of row row0[k] + li: the tile goes back to LDS as
// one conflict-free ds_write_b128 per lane and unit instead of four ds_write_b32 (64 B/clk/CU; the H and Z write-backs
// were ~1 k cycles per stage each).  Fragments of k block q+1 are requested before the MFMAs of block q.
// kmask >= 0: the rows of A are stored with their 16-B chunks XOR-swizzled by (row & kmask) (row0 multiples of 16, so the
// key of a lane's row is li & kmask): chunk 4 q + lg of the row sits at 4 (q ^ kq) + (lg ^ (li & 3)), kq = (li >> 2) & (kmask >> 2)
template <int KQ, int NU>
__device__ __forceinline__ void zf_mma(const float *__restrict__ A, int lda, const float (&wr)[KQ * 4], const int (&row0)[NU],
                                       int li, int lg, f32x4 (&acc)[NU], int nt = 4, int kmask = -1)
{
    const float *ap[NU];
    const int lgs = kmask >= 0 ? (lg ^ (li & 3 & kmask)) : lg;
    const int kq = kmask >= 0 ? ((li >> 2) & (kmask >> 2)) : 0;
#pragma unroll
    for (int k = 0; k < NU; k++)
        ap[k] = A + (row0[k] + li) * lda + lgs * 4;
    float4 a4[NU], an[NU];
#pragma unroll
    for (int k = 0; k < NU; k++)
        a4[k] = *reinterpret_cast<const float4 *>(ap[k] + 16 * kq);
#pragma unroll
    for (int q = 0; q < KQ; q++) {
        if (q + 1 < KQ) {
#pragma unroll
            for (int k = 0; k < NU; k++)
                an[k] = *reinterpret_cast<const float4 *>(ap[k] + 16 * ((q + 1) ^ kq));
        }
#if ZF_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t >= nt) // (wave-uniform; nt = 3: the block's fourth k step holds zeros on both sides -- narrow inputs, ZF_K12)
                break;
#pragma unroll
            for (int k = 0; k < NU; k++) {
                const float av = t == 0 ? a4[k].x : (t == 1 ? a4[k].y : (t == 2 ? a4[k].z : a4[k].w));
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q * 4 + t], av, acc[k], 0, 0, 0);
            }
        }
#if ZF_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (q + 1 < KQ) {
#pragma unroll
            for (int k = 0; k < NU; k++)
                a4[k] = an[k];
        }
    }
}

// ---- MX = 1 (opt-in, gnnb_set_option("math", 2), "bf16x3"): M1 on the bf16 matrix cores with BOTH operands as hi + mid bf16
// pieces (round to nearest even: |x - hi - mid| <= 2^-17 |x|) and the three products hi.hi + hi.mid + mid.hi, fp32 accumulate:
// 3 v_mfma_f32_16x16x32_bf16 (4 passes each) per 32-wide k block instead of 8 fp32 MFMAs of 8 passes -- 5.3x fewer matrix-pipe
// cycles -- at ~18 significant bits per product (between tf32's 11 and fp32's 24).  A REDUCED-PRECISION study mode, never the
// default and never bench.py's `value` (the analogue of the reference's float_or_fixed switch, code_gen.py:39-52).
// H is written by M0 as two bf16 planes inside the SAME row the fp32 form uses ([hi: h0 x 2 B][mid: h0 x 2 B][pad]): no LDS
// more, Z goes back over it in fp32 as before.  (The fp32-equivalent bf16x6 form needs a third plane, 6 B per element: the
// 176-row stage would not fit, and 144-row stages turn two stages per CU at BASELINE config 2 into three.)
__device__ __forceinline__ uint32_t bf16_rne(float x) // the bf16 nearest to x, as an fp32 bit pattern (low half zero)
{
    uint32_t u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u & 0xffff0000u;
}
__device__ __forceinline__ void split2(float x, uint32_t &h, uint32_t &m)
{
    h = bf16_rne(x);
    m = bf16_rne(x - __uint_as_float(h)); // (the difference is exact in fp32)
}
// MX = 2 ("f16x3"): the same with fp16 pieces (v_cvt_f16_f32 rounds to nearest even; x - hi is exact in fp32): 11 + 11 significant
// bits, |x - hi - mid| <= 2^-22 |x| -- sixteen times closer than the bf16 pieces at the same cost -- but fp16's RANGE: values of
// 65520 and above become inf, and pieces below 6e-8 are lost (an absolute floor of 3e-8 per operand element).
// four consecutive fp32 values -> their hi pieces and their mid pieces, two 16-bit pieces per dword
template <int MX>
__device__ __forceinline__ void split2x4(const float4 &v, uint2 &hi, uint2 &mid)
{
    if constexpr (MX == 2) {
        const float x[4] = {v.x, v.y, v.z, v.w};
        _Float16 h[4], m[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            h[i] = (_Float16)x[i];
            m[i] = (_Float16)(x[i] - (float)h[i]);
        }
        hi = make_uint2(pack_f16(h[0], h[1]), pack_f16(h[2], h[3]));
        mid = make_uint2(pack_f16(m[0], m[1]), pack_f16(m[2], m[3]));
    } else {
        uint32_t h0, m0, h1, m1, h2, m2, h3, m3;
        split2(v.x, h0, m0);
        split2(v.y, h1, m1);
        split2(v.z, h2, m2);
        split2(v.w, h3, m3);
        hi = make_uint2(pack_hi16(h0, h1), pack_hi16(h2, h3));
        mid = make_uint2(pack_hi16(m0, m1), pack_hi16(m2, m3));
    }
}
template <int MX>
__device__ __forceinline__ f32x4 mfma_16x3(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (MX == 2) {
        union {
            u32x4 u;
            f16x8 h;
        } ca, cb;
        ca.u = a;
        cb.u = b;
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(ca.h, cb.h, c, 0, 0, 0);
    } else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
}
// Bank conflicts: this form of M1 is bound by the LDS array, so its fragment reads must be conflict-free.  A ds_read_b128 is
// served in four groups of sixteen lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32); with rows 33 slots of
// 16 B apart lane (li, lg) of the plain layout reads slot li + lg (mod 16), and rows {12..15} at chunk lg meet rows {4..11} at
// chunk lg + 1.  The 16-B chunks of rows 4..11 (mod 16) are therefore stored with the lowest bit of their index flipped
// (key 1): the four groups then read slots {0-3, 12-15} u {4-11}, {5-12} u {1-4, 13-16}, {2-5, 14-17} u {6-13},
// {7-14} u {3-6, 15-18} -- sixteen different ones each.  The key is a per-lane constant on both sides (M0 writes 8 B of a chunk).
__device__ __forceinline__ int zf_bf_key(int li) { return ((li + 4) >> 3) & 1; }
// acc[k] += Wslice . A[rows of unit k][:]^T over K = 32 KQ32, transposed tile as zf_mma.  wr: per 32-wide k block q the lane's
// eight k values 32 q + 8 lg .. + 7 of its weight row as {hi x 4 dwords, mid x 4 dwords}; Hb: rows of two bf16 planes (the mid
// plane `midoff` bytes behind the hi plane), lane (li, lg) reads 16 B of each plane at k = 32 q + 8 lg.
template <int MX, int KQ32, int NU>
__device__ __forceinline__ void zf_mma_bf3(const char *__restrict__ Hb, int ldhb, int midoff, const float (&wr)[KQ32 * 8], const int (&row0)[NU],
                                           int li, int lg, f32x4 (&acc)[NU])
{
    const char *ap[NU];
    {
        int lane_off = li * ldhb + (lg ^ zf_bf_key(li)) * 16;
        asm volatile("" : "+v"(lane_off)); // (opaque: left alone the compiler keeps the pieces of this sum live through the MFMA loop)
#pragma unroll
        for (int k = 0; k < NU; k++)
            ap[k] = Hb + row0[k] * ldhb + lane_off;
    }
    // ONE fragment buffer (the fp32 form keeps two): block q + 1's fragments are requested BEHIND block q's MFMAs, into the
    // same registers -- 24 instead of 48, which keeps the kernel inside the 104-register budget the other batches' guest
    // kernels depend on; the LDS round trip is covered by the other three waves of the SIMD, all of them in M1 (the phase is
    // bound by the LDS array in this form: every wave reads its units' rows for ONE 16-column slice, 720 KB per 176-row stage
    // = 5.6 k cycles at 128 B per clock, against 3.5 k cycles of matrix time)
    u32x4 ah[NU], am[NU];
#pragma unroll
    for (int k = 0; k < NU; k++) {
        ah[k] = *reinterpret_cast<const u32x4 *>(ap[k]);
        am[k] = *reinterpret_cast<const u32x4 *>(ap[k] + midoff);
    }
#pragma unroll
    for (int q = 0; q < KQ32; q++) {
#if ZF_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
        const u32x4 wh = {__float_as_uint(wr[q * 8 + 0]), __float_as_uint(wr[q * 8 + 1]), __float_as_uint(wr[q * 8 + 2]), __float_as_uint(wr[q * 8 + 3])};
        const u32x4 wm = {__float_as_uint(wr[q * 8 + 4]), __float_as_uint(wr[q * 8 + 5]), __float_as_uint(wr[q * 8 + 6]), __float_as_uint(wr[q * 8 + 7])};
        // (the two small products first, then the large one; unit by unit inside a product: consecutive MFMAs never share an accumulator)
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = mfma_16x3<MX>(wh, am[k], acc[k]);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = mfma_16x3<MX>(wm, ah[k], acc[k]);
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = mfma_16x3<MX>(wh, ah[k], acc[k]);
#if ZF_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (q + 1 < KQ32) {
#pragma unroll
            for (int k = 0; k < NU; k++) {
                am[k] = *reinterpret_cast<const u32x4 *>(ap[k] + midoff + 64 * (q + 1));
                ah[k] = *reinterpret_cast<const u32x4 *>(ap[k] + 64 * (q + 1));
            }
        }
    }
}

struct ZfStage {
    int ok, chunk, nb, rows, ga, gb, e0, ne; // ok = 0: no stage (the hand-out is exhausted)
};

// REGISTER BUDGET: keep this kernel at <= 104 VGPRs (`make resource-usage`).  Four waves per SIMD then leave a 96-register
// wave slot, and with ~145 KB of LDS per CU that is what lets the readout and graph-prep kernels of the other batches in
// flight run BESIDE it: at 111 registers the three-stream pipeline of bench.py lost 12 % (59.0 vs 52.1 us per step).
// (The launch bound only promises four waves per SIMD = 128 registers; amdgpu_num_vgpr is ignored beside it.)
// Round 6: the forms below that fit it take 96 (a launch bound of FIVE waves per SIMD): the hole beside four of them is then 128
// registers -- two guest waves per SIMD.
// (round 6) HEAD = false: no MLP-head tail in the kernel's text, and for the fp32 form a budget of 96 registers (five waves per
// SIMD by the launch bound: 99 -> 95-96 without a spill once the tail is gone).  Beside four such waves a SIMD has 128 registers
// left: two guest waves (graph prep 56, or prep + ... the readout's 88 alone) instead of one -- the driver's 20-step region
// 46.2 -> 45.1 us per step (six alternating runs each), `--steps 200` and the kernel alone unchanged (37.3 vs 37.2 us).
// Which instantiations take the 96-register budget: the ones that fit it WITHOUT a spill (`make resource-usage`; the CPU test
// test_no_stack_kernel_spills_to_scratch holds every instantiation to zero scratch) -- the fp32 forms without the head tail at
// hidden widths 32 / 64, and at hidden 128 the ReLU form with a one-block input (every BASELINE GCN model); hidden 128 with
// GELU / sigmoid / tanh or a two-block input needs 97-100 and keeps the 104-register budget.
template <int ACT, int KQ0, int KQ1, int MX, bool HEAD>
constexpr int zf_waves_per_simd()
{
    return (!HEAD && MX == 0 && (KQ1 < 8 || (ACT == GNNB_ACT_RELU && KQ0 == 1))) ? 5 : 4;
}
template <int ACT, int KQ0, int KQ1, int NW, int ZF_UNITS, int MX = 0, bool H1FULL = false, bool HEAD = true>
__global__ __launch_bounds__(NW * 64, (zf_waves_per_simd<ACT, KQ0, KQ1, MX, HEAD>())) void k_gcn2_zf(
    const float *__restrict__ x, int f0, const int4 *__restrict__ node_rec,
    const int32_t *__restrict__ col, const float *__restrict__ dinv,
    const int32_t *__restrict__ tile_first, const int32_t *__restrict__ tile_graph, const int32_t *__restrict__ tile_edge,
    const int32_t *__restrict__ node_ptr, int num_tiles, int num_graphs, int N, int E, const float *__restrict__ W0,
    const float *__restrict__ b0, int h0, const float *__restrict__ W1, const float *__restrict__ W1f,
    const float *__restrict__ b1, int h1, int p0, int p1, int p2, int np, float *__restrict__ pooled,
    const HeadArgs *__restrict__ head_dev, float *__restrict__ head_out, int head_ldact, // head_dev != nullptr: the MLP head runs here too (round 5, below)
    int32_t *__restrict__ err, int32_t *__restrict__ err_host // MX != 0: the workspace's flag word (GNNB_FLAG_RANGE: a non-finite Z, gnnb_device.h RangeProbe)
#ifdef GNNB_ZF_ABLATE
    , unsigned long long *dbg_span // [2]: min start / max end wall clock (100 MHz) over the workgroups of this launch
    , int dbg // development only (-DGNNB_ZF_ABLATE): bit 0 skips P1, 1 skips P0', 2 skips M1, 3 skips M0, 4 skips the Z write
#define ZF_ON(bit) (!(dbg & (1 << (bit))))
#else
#define ZF_ON(bit) true
#endif
)
{
    // WIDTHS THE COMPILER MAY TREAT AS CONSTANTS (round 6; same-box A/B at BASELINE config 2, wide shape, us per launch: all run-time
    // 38.7; h0 38.25; + layer 1's wave roles 37.95; + P1's lane geometry 37.3; + the H / Z row stride or the Z write's column
    // predicate +-0; h1 constant EVERYWHERE -- prologue loads, carve, stride -- 42.3: slower than none, as round 4 found, "the
    // kernel sits in a code-generation optimum that instruction counts do not predict").  h0 is 16 KQ1 by dispatch; H1FULL says
    // the last layer is as wide (h1 == 16 KQ1: every BASELINE GCN model): the wave roles (cs1l, nrg1) and P1's (csl, Gl, S, the lane
    // -> (row slot, chunk) map) are then literals -- ~45 scalar registers less to spill into vector lanes and read back with
    // v_readlane inside the stage loop (the ISA had 92 such instructions there, 27 now), and the two scalar loops that re-derived
    // them per stage are gone.  h1 itself stays the run-time argument everywhere else.
    h0 = 16 * KQ1;
    const int h1g = H1FULL ? 16 * KQ1 : h1; // layer 1's wave roles
    const int h1p = H1FULL ? 16 * KQ1 : h1; // P1's lane geometry and the pooled row's width
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ZF_CAP = 16 * ZF_UNITS, G2_NW = NW;
    constexpr int GMAX = ZF_CAP <= 96 ? 64 : 128; // graph boundaries of a stage kept in LDS (more: empty graphs piling up)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- LDS carve (bytes, every region 16-B aligned; LDS pointers are always derived arithmetically from `smem`:
    // runtime-indexed arrays of LDS pointers turn into FLAT accesses, see k_stack.hip)
    //   ROWS   xs | srec                     ONE buffer (P0 is its only reader; refilled one stage ahead)
    //   SMALL  sdinv | node_ptr of <= 64 graphs (+ end)      TWO buffers
    //   A0     [CAP][LD0]                    P0 -> M0
    //   H      [CAP][ldh]                    M0 -> M1, then Z in place -> P1
    //   REC    [CAP] x 48 B                  TWO buffers (P0 of the next stage writes while P1 of this one reads)
    //   SCOL   [ECAP] int32                  TWO buffers: the stage's slice of the CSR `col` array, for rows of degree > 4
    //                                        (a tracked global read there costs a full memory round trip per neighbour)
    //   SB1, SB0, SPLAN                      biases, the plan of the stage after next
    constexpr int LD0 = 16 * KQ0; // A0 row: F0 values zero-padded to whole 16-wide MFMA k blocks
    const int xs_b = ((ZF_CAP * f0 * 4) + 15) & ~15;
    const int rows_b = xs_b + ZF_CAP * 32;
    constexpr int small_b = ZF_CAP * 4 + ((GMAX + 1) * 4 + 15) / 16 * 16;
    constexpr int a0_b = ZF_CAP * LD0 * 4;
#if ZF_SWZ
    // H / Z rows: 32, 64 or 128 floats, NOT padded; the 16-B chunks of row r are stored XOR-swizzled by r & kmask.  A
    // ds_read_b128 is served in four groups of sixteen lanes ({0-3, 12-15, 20-27}, ...): the fragment reads of M0 / M1 put
    // rows {-4..3} at chunk lg and rows {4..11} at chunk lg + 1 into one group -- with a padded row (33 slots) two of the
    // sixteen always met on one slot (a third of the kernel's LDS cycles are conflicts, most of them here); XOR by the row index maps the two
    // row sets onto disjoint slot sets whatever the chunk.  The region starts on a 512-B boundary so that a row's base
    // and its key do not share bits: P1 forms a neighbour's address as (record offset) ^ (the lane's chunk << 4).
    const int ldh = (h0 > 64 || h1 > 64) ? 128 : ((h0 > 32 || h1 > 32) ? 64 : 32);
    const int kmask = (ldh >= 64 ? 16 : 8) - 1;
    const int hoff = (rows_b + 2 * small_b + a0_b + 511) & ~511;
#else
    const int ldh = (h0 > h1 ? h0 : h1) + 4; // padded H / Z row (floats): conflict-free fragment reads, base + immediate
    const int kmask = -1;
    const int hoff = rows_b + 2 * small_b + a0_b;
#endif
    const int ldhb = ldh * 4;
    auto hswz = [&](int row) { return ZF_SWZ ? ((row & kmask) << 4) : 0; }; // a row's key, as a byte offset
    constexpr int rec_b = ZF_CAP * 48;
    float *A0 = reinterpret_cast<float *>(smem + rows_b + 2 * small_b);
    float *H = reinterpret_cast<float *>(smem + hoff);
    char *RECb = reinterpret_cast<char *>(H) + ZF_CAP * ldhb;
    constexpr int ECAP = ZF_CAP <= 96 ? 512 : 1024;
    char *SCOLb = RECb + 2 * rec_b;
    float *SB1 = reinterpret_cast<float *>(SCOLb + 2 * ECAP * 4); // b1 zero-padded to 128 floats
    int4 *SPLAN = reinterpret_cast<int4 *>(SB1 + 128);                 // the stage after next, planned by ONE wave (2 x int4)
    float *SB0 = reinterpret_cast<float *>(SPLAN + 2);                 // b0 zero-padded to 128 floats
    int *STAB = reinterpret_cast<int *>(SB0 + 128);                    // the planner wave's copy of the run's tile-table entries (3 x 64)

#ifdef GNNB_ZF_ABLATE
    if (dbg_span && threadIdx.x == 0)
        atomicMin(dbg_span, wall_clock64());
    struct SpanEnd {
        unsigned long long *p;
        __device__ ~SpanEnd() { if (p && threadIdx.x == 0) atomicMax(p + 1, wall_clock64()); }
    } span_end{dbg_span};
    if (dbg & 32)
        return; // (launch overhead alone)
    // de-phasing experiment: bits 8..15 = delay in units of 0.25 us for half of the workgroups, bits 16..17 = which half
    // (0: the second half of the grid, 1: odd block ids, 2: odd groups of eight, 3: odd groups of 256/8)
    if ((dbg >> 8) & 255) {
        const int sel = (dbg >> 16) & 3, b_ = blockIdx.x;
        const bool mine = sel == 0 ? b_ >= (int)gridDim.x / 2 : (sel == 1 ? (b_ & 1) : (sel == 2 ? ((b_ >> 3) & 1) : ((b_ >> 8) & 1)));
        if (mine) {
            const unsigned long long t_ = wall_clock64(), dt = 25ull * ((dbg >> 8) & 255);
            while (wall_clock64() - t_ < dt)
                __builtin_amdgcn_s_sleep(16);
        }
    }
#endif
    // ---- the workgroup's run of node tiles: equal tile counts (= equal rows up to one graph).  (A ticket hand-out of
    // fixed-size chunks was built and measured: a chunk must fit a stage whatever its last graph's overhang, i.e. 64
    // nominal rows of a 96-row stage, which turns two stages per workgroup into 2.25 -- three rounds, 55 us instead of 44.
    // The planner below uses the capacity adaptively instead: what one stage's overhang takes the other gives.)
#ifdef ZF_OLD_CUTS // (development A/B: the two 64-bit software divisions every wave ran until round 6)
    const int t0 = (int)(((long long)blockIdx.x * num_tiles) / gridDim.x);
    const int t1 = (int)(((long long)(blockIdx.x + 1) * num_tiles) / gridDim.x);
#else
    int t0, t1;
    run_cuts(blockIdx.x, gridDim.x, (unsigned)num_tiles, t0, t1); // (32-bit: gnnb_device.h)
#endif
    if (t1 <= t0)
        return;
    // Every wave fetches the run's tile-table entries into REGISTERS (lane l: tile t0 + l; the launcher keeps runs below
    // 64 tiles) and plans the first stage with v_readlane: no LDS copy to wait for, no barrier in front of the first DMA.
    // The registers are short-lived (holding them through the stage loop cost 8 VGPRs and with them the 96-register wave
    // slot the other batches' kernels run in): the planner wave parks its copy in LDS and reloads it where it plans.
    // (clamped: the tables of a malformed batch may hold stale entries; a flagged batch must still stay in range)
    int tf, tg, te;
    {
        const int ti = min(t0 + min(lane, t1 - t0), num_tiles);
        tf = min(max(tile_first[ti], 0), N);
        tg = min(max(tile_graph[ti], 0), num_graphs);
        te = min(max(tile_edge[ti], 0), E);
    }
    if (wave == NW - 1) {
        STAB[lane] = tf;
        STAB[64 + lane] = tg;
        STAB[128 + lane] = te;
    }
    // ---- wave roles: layer L has ncs_L = pow2ceil(h_L / 16) column slices of 16 and nrg_L = 8 / ncs_L row groups;
    // wave w owns slice (w mod ncs) for the units rg, rg + nrg, ... with rg = w / ncs
    int cs0l = 0, cs1l = 0;
    while ((16 << cs0l) < h0)
        cs0l++;
    while ((16 << cs1l) < h1g)
        cs1l++; // h <= 128 -> <= 3
    const int nrg0 = G2_NW >> cs0l, nrg1 = G2_NW >> cs1l;
    constexpr int LOG2NW = NW == 16 ? 4 : 3;
    static_assert(NW == 8 || NW == 16, "wave roles assume 8 or 16 waves");
    const int lnrg0 = LOG2NW - cs0l, lnrg1 = LOG2NW - cs1l;


    // ---- balanced stage plan: the rows that are left are cut into the fewest stages that can hold them, of EQUAL
    // size, at tile (= graph) boundaries.  A stage takes the boundary closest to its share; boundaries that would
    // leave more than the remaining stages can hold are only taken when there is no other (then the largest).
    // The search runs ACROSS the lanes (lane l holds tile t0 + l): the boundaries that fit a stage are a prefix of the
    // lanes behind `ta`, their row counts ascend, so the best cut is the last one below the share or the first one at or
    // above it -- two ballots and a comparison instead of a loop over the candidates (that loop, ~100 instructions, ran in
    // every wave in front of the first DMA and in the planner wave every stage).
    auto plan = [&](int ta, int tf, int tg, int te, int ln) { // ln = this lane's index
        auto T_first = [&](int t) { return __builtin_amdgcn_readlane(tf, __builtin_amdgcn_readfirstlane(t - t0)); };
        auto T_graph = [&](int t) { return __builtin_amdgcn_readlane(tg, __builtin_amdgcn_readfirstlane(t - t0)); };
        auto T_edge = [&](int t) { return __builtin_amdgcn_readlane(te, __builtin_amdgcn_readfirstlane(t - t0)); };
        ZfStage st;
        st.ok = ta < t1;
        st.chunk = ta;
        st.nb = st.rows = st.ga = st.gb = st.e0 = st.ne = 0;
        if (!st.ok)
            return st;
        st.nb = T_first(ta);
        const int rrem = T_first(t1) - st.nb;
        const int krem = max((rrem + ZF_CAP - 1) / ZF_CAP, 1);
        const int target = (rrem + krem - 1) / krem;
        const int rmin = rrem - (krem - 1) * ZF_CAP;
        const int rel = ta - t0;
        const int r = tf - st.nb; // rows of a stage that ends at this lane's tile
        const unsigned long long feas = __ballot(ln > rel && ln <= t1 - t0 && r <= ZF_CAP);
        const unsigned long long ge = feas & __ballot(r >= target);
        const unsigned long long lt = feas & ~ge;
        int pick = rel + 1; // (nothing fits: the next tile alone, only if the max_graph_nodes promise is broken)
        if (feas) {
            const int hi = ge ? __builtin_ctzll(ge) : -1, lo = lt ? 63 - __builtin_clzll(lt) : -1;
            if (hi < 0)
                pick = lo;
            else if (lo < 0)
                pick = hi;
            else {
                const int r_lo = __builtin_amdgcn_readlane(r, __builtin_amdgcn_readfirstlane(lo)), r_hi = __builtin_amdgcn_readlane(r, __builtin_amdgcn_readfirstlane(hi));
                const int d_lo = r_lo < rmin ? 4096 + (rmin - r_lo) : target - r_lo, d_hi = r_hi - target;
                pick = d_hi <= d_lo ? hi : lo;
            }
            // (ties: the LAST boundary with the same row count, so that empty tiles are swallowed)
            const unsigned long long same = feas & __ballot(r == __builtin_amdgcn_readlane(r, __builtin_amdgcn_readfirstlane(pick)));
            pick = 63 - __builtin_clzll(same);
        }
        const int tb = t0 + pick;
        st.chunk = tb; // (the next stage starts here)
        st.rows = max(min(T_first(tb) - st.nb, ZF_CAP), 0); // (> CAP only if the max_graph_nodes promise is broken)
        st.ga = T_graph(ta);
        // (empty graphs after the last node belong to the last stage: when N is a multiple of the tile
        // size the first of them already owns tile_graph[num_tiles])
        st.gb = max(tb == num_tiles ? num_graphs : T_graph(tb), st.ga);
        st.e0 = T_edge(ta);
        st.ne = max(T_edge(tb) - st.e0, 0);
        return st;
    };
    // the stage's rows (x, node records) -> ROWS
    // (round 4: the same bytes as 16-B LDS-DMA pieces dealt one per wave -- 16 instructions per stage instead of ~40, the
    // global side of global_load_lds_dwordx4 takes dword-aligned addresses -- are SLOWER, 40.4 vs 39.9 us: the issue phase
    // grew from 0.9-2.2 k to 1.2-3.6 k cycles per wave; misaligned 16-B pieces cost the issuing wave more than four dword ones)
    auto issue_rows = [&](const ZfStage &st, int bb, int lane, int wave) {
        if (!st.ok)
            return;
        dma_dwords_u(x + (size_t)st.nb * f0, smem, st.rows * f0, wave, lane, G2_NW);
        if (st.ne <= ECAP) // (a stage with more edges -- hubs, multigraphs -- reads `col` from global memory)
            dma_dwords_u(col + st.e0, SCOLb + (size_t)bb * ECAP * 4, st.ne, (wave + G2_NW / 2) & (G2_NW - 1), lane, G2_NW);
        const char *grec = reinterpret_cast<const char *>(node_rec + 2 * (size_t)st.nb);
        const int rbytes = st.rows * 32; // 1 KiB per wave: waves 0 .. CAP / 32 - 1
        if (wave * 1024 + lane * 16 < rbytes)
            dma16_to_lds_u(grec + wave * 1024 + lane * 16, smem + xs_b + wave * 1024);
    };
    // its normalisers and graph boundaries -> SMALL[bb]
    auto issue_small = [&](const ZfStage &st, int bb, int lane, int wave) {
        if (!st.ok)
            return;
        // (a stage may have NO rows and still own graphs: empty graphs behind a graph that ends on the
        // tile edge -- their boundaries are still needed by the pooling)
        char *base = smem + rows_b + (size_t)bb * small_b;
        const int ng = min(st.gb - st.ga, GMAX) + 1;
        // 64-dword pieces, one per wave from wave 3 on: ND pieces of dinv, then NG pieces of the graph boundaries of
        // the stage (first GMAX graphs; more only if empty graphs pile up, those are read from global memory)
        constexpr int ND = (ZF_CAP + 63) / 64, NG = (GMAX + 1 + 63) / 64;
        static_assert(3 + ND + NG <= NW, "one small-DMA piece per wave");
        const int pc = wave - 3;
        if (pc >= 0 && pc < ND) {
            if (pc * 64 + lane < st.rows)
                dma4_to_lds_u(dinv + st.nb + pc * 64 + lane, base + pc * 256);
        } else if (pc >= ND && pc < ND + NG) {
            const int o = (pc - ND) * 64;
            if (o + lane < ng)
                dma4_to_lds_u(node_ptr + st.ga + o + lane, base + ZF_CAP * 4 + o * 4);
        }
    };

    // the first stage's inputs start their way to LDS before the weights are fetched (both are waited for below)
    // (every wave plans the first stage for itself from its registers)
    ZfStage cur = plan(t0, tf, tg, te, lane);
    issue_small(cur, 0, lane, wave);
    issue_rows(cur, 0, lane, wave);
    // (the biases -> LDS behind the first DMA: tracked loads, nobody reads them before the two barriers that close the
    // prologue; in front of the plan they held waves 0 and 1 back for a memory round trip)
    if (tid < 128) {
        SB1[tid] = (b1 && tid < h1) ? b1[tid] : 0.0f;
        SB0[tid] = (b0 && tid < h0) ? b0[tid] : 0.0f;
    }
    // the second stage: planned by the last wave, handed over through LDS behind the barrier that closes the prologue's P0
    auto publish = [&](const ZfStage &pn, int lane) {
        if (lane == 0) {
            SPLAN[0] = make_int4(pn.ok, pn.chunk, pn.nb, pn.rows);
            SPLAN[1] = make_int4(pn.ga, pn.gb, pn.e0, pn.ne);
        }
    };
    if (wave == G2_NW - 1)
        publish(plan(cur.chunk, tf, tg, te, lane), lane);

    // (the weights are requested HERE, behind the first stage's DMA: in front of the tile-table loads they made the
    // workgroup's first barrier wait for 128 KB of weight fragments; now they land beside the DMA round trip and P0)
    // ---- weight slices -> registers (16 output columns x K per layer and wave), biases
    float w0r[KQ0 * 4], w1r[KQ1 * 4];
    {
        const int li = lane & 15, lg = lane >> 4;
        const int n0c = (wave & ((1 << cs0l) - 1)) * 16 + li, n1c = (wave & ((1 << cs1l) - 1)) * 16 + li;
#pragma unroll
        for (int q = 0; q < KQ0; q++) {
            [[maybe_unused]] const int k = 16 * q + 4 * lg;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#if ZF_K12
            // k step t of block q multiplies input feature 16 q + lg + 4 t (A0 is stored to match, phase_p0): a model of
            // up to 12 input features (QM9: 11) leaves the fourth step of its only block empty -- M0 skips it
            if (n0c < h0) {
                const float *wrow = W0 + (size_t)n0c * f0;
                const int kk = 16 * q + lg;
                v.x = kk < f0 ? wrow[kk] : 0.f;
                v.y = kk + 4 < f0 ? wrow[kk + 4] : 0.f;
                v.z = kk + 8 < f0 ? wrow[kk + 8] : 0.f;
                v.w = kk + 12 < f0 ? wrow[kk + 12] : 0.f;
            }
#else
            if (n0c < h0)
                v = load4_guard(W0 + (size_t)n0c * f0 + k, f0 - k, false);
#endif
            w0r[q * 4 + 0] = v.x;
            w0r[q * 4 + 1] = v.y;
            w0r[q * 4 + 2] = v.z;
            w0r[q * 4 + 3] = v.w;
        }
        if constexpr (MX != 0) {
            // (bf16x3: per 32-wide k block the lane's eight k values 32 q + 8 lg .. + 7 of weight row n1c, split into hi and
            // mid bf16 pieces HERE -- once per workgroup, ~100 instructions -- and kept in the same 4 KQ1 registers)
            static_assert(KQ1 % 2 == 0 && !ZF_SWZ, "bf16x3: whole 32-wide k blocks, padded rows");
#pragma unroll
            for (int q = 0; q < KQ1 / 2; q++) {
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (n1c < h1) {
                    v0 = *reinterpret_cast<const float4 *>(W1 + (size_t)n1c * h0 + 32 * q + 8 * lg);
                    v1 = *reinterpret_cast<const float4 *>(W1 + (size_t)n1c * h0 + 32 * q + 8 * lg + 4);
                }
                uint2 ha, ma, hb, mb;
                split2x4<MX>(v0, ha, ma);
                split2x4<MX>(v1, hb, mb);
                w1r[q * 8 + 0] = __uint_as_float(ha.x);
                w1r[q * 8 + 1] = __uint_as_float(ha.y);
                w1r[q * 8 + 2] = __uint_as_float(hb.x);
                w1r[q * 8 + 3] = __uint_as_float(hb.y);
                w1r[q * 8 + 4] = __uint_as_float(ma.x);
                w1r[q * 8 + 5] = __uint_as_float(ma.y);
                w1r[q * 8 + 6] = __uint_as_float(mb.x);
                w1r[q * 8 + 7] = __uint_as_float(mb.y);
            }
        } else {
#pragma unroll
        for (int q = 0; q < KQ1; q++) {
            const int k = 16 * q + 4 * lg; // h0 == 16 * KQ1
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (W1f) // fragment-order copy (gnnb_model_create): one contiguous KiB per load instruction of the wave
                v = reinterpret_cast<const float4 *>(W1f)[(((wave & ((1 << cs1l) - 1)) * KQ1 + q) * 4 + lg) * 16 + li];
            else if (n1c < h1)
                v = *reinterpret_cast<const float4 *>(W1 + (size_t)n1c * h0 + k);
            w1r[q * 4 + 0] = v.x;
            w1r[q * 4 + 1] = v.y;
            w1r[q * 4 + 2] = v.z;
            w1r[q * 4 + 3] = v.w;
        }
        }
    }

    // Pin every weight register through an (empty) asm: the compiler must finish the loads HERE (k_stack.hip: left
    // alone it guards their first use inside the stage loop with s_waitcnt vmcnt(0), which also waits for the DMA)
#pragma unroll
    for (int q = 0; q < KQ0 * 4; q++)
        asm volatile("" : "+v"(w0r[q]));
    dma_wait_all();
    __syncthreads();

    const int pools[3] = {p0, p1, p2};

    // ---- P0: A0[i][f] = sum_j x_j[f] dinv_i dinv_j + x_i[f] dinv_i^2   (CSR order, self last) and the per-row record
    // for P1: {byte offsets of the 4 inline neighbour rows in H}{coefficients dinv_i dinv_j, 0 past the degree}
    // {dinv_i^2, rp0, deg, dinv_i}.  Eight lanes per row, lane l8 takes features l8, l8 + 8, ...; a wave pass = 8 rows;
    // wave-pass p of the stage is done by wave (pstart + p) mod 8.  Every LDS load is unconditional (unused neighbour
    // slots alias the row itself, inactive lanes read row 0) and the degree only selects.
    auto phase_p0 = [&](const ZfStage &st, int bb, int tv, int pstart) {
        constexpr int T0 = LD0 / 8;
        const float *xs = reinterpret_cast<const float *>(smem);
        const int4 *srec = reinterpret_cast<const int4 *>(smem + xs_b);
        const float *sdinv = reinterpret_cast<const float *>(smem + rows_b + (size_t)bb * small_b);
        int4 *REC = reinterpret_cast<int4 *>(RECb + (size_t)bb * rec_b);
        const int32_t *scol = reinterpret_cast<const int32_t *>(SCOLb + (size_t)bb * ECAP * 4);
        const bool col_lds = st.ne <= ECAP;
        const int e0 = st.e0;
        const int wv = __builtin_amdgcn_readfirstlane(tv >> 6), l8 = tv & 7, r8 = (tv >> 3) & 7;
        const int rows = st.rows, nb = st.nb;
        const int npass = (rows + 7) >> 3;
        for (int p = (wv - pstart) & (G2_NW - 1); p < npass; p += G2_NW) {
            const int i = p * 8 + r8;
            const bool active = i < rows;
            const int ic = active ? i : 0;
            const int4 r0 = srec[2 * ic], r1 = srec[2 * ic + 1];
            const int deg = r0.y;
            const int jl[4] = {r0.z - nb, r0.w - nb, r1.x - nb, r1.y - nb};
            const float di = sdinv[ic];
            float xv[T0][4], xself[T0], sv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                sv[q] = sdinv[jl[q]];
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
                // degree > 4: the rest of the CSR row, from the stage's slice in LDS (two loops, not a select between an
                // LDS and a global pointer: that becomes a flat load with a full drain)
                auto more = [&](int j) {
                    const float cj = di * sdinv[j];
#pragma unroll
                    for (int t = 0; t < T0; t++) {
                        const int f = l8 + 8 * t;
                        acc[t] += xs[j * f0 + (f < f0 ? f : 0)] * cj;
                    }
                };
                if (col_lds) {
                    for (int k = r0.x + 4; k < r0.x + deg; k++)
                        more(scol[min(max(k - e0, 0), ECAP - 1)] - nb);
                } else {
                    for (int k = r0.x + 4; k < r0.x + deg; k++)
                        more(col[k] - nb);
                }
#pragma unroll
                for (int t = 0; t < T0; t++) {
                    const int f = l8 + 8 * t;
#if ZF_K12
                    const int fp = (f & ~15) | ((f & 3) << 2) | ((f >> 2) & 3); // (feature lg + 4 t sits at position 4 lg + t of its block)
#else
                    const int fp = f;
#endif
                    A0[i * LD0 + fp] = f < f0 ? acc[t] + xself[t] * (di * di) : 0.0f;
                }
                if (l8 == 0) {
#if ZF_SWZ
                    // (LDS byte address of the neighbour's row + its swizzle key: P1 XORs its chunk offset in)
                    REC[3 * i] = make_int4(hoff + jl[0] * ldhb + hswz(jl[0]), hoff + jl[1] * ldhb + hswz(jl[1]),
                                           hoff + jl[2] * ldhb + hswz(jl[2]), hoff + jl[3] * ldhb + hswz(jl[3]));
#else
                    REC[3 * i] = make_int4(jl[0] * ldhb, jl[1] * ldhb, jl[2] * ldhb, jl[3] * ldhb);
#endif
                    REC[3 * i + 1] = make_int4(__float_as_int(c[0]), __float_as_int(c[1]), __float_as_int(c[2]), __float_as_int(c[3]));
                    REC[3 * i + 2] = make_int4(__float_as_int(di * di), r0.x, deg, __float_as_int(di));
                }
            }
        }
    };

#ifdef GNNB_PROBE
    unsigned long long pt[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = clock64(), pw0 = wall_clock64(), pt_last = pt0;
    int nst = 0;
    unsigned long long prows = 0, pgraphs = 0, punits = 0;
#define ZF_PT(i) do { const unsigned long long _n = clock64(); pt[i] += _n - pt_last; pt_last = _n; } while (0)
#elif defined(GNNB_ZF_MARK) // static instruction table (tools/isa_table.py): phase boundaries as comments in the ISA
#define ZF_PT(i) asm volatile("; ZFMARK " #i)
#else
#define ZF_PT(i) do { } while (0)
#endif

#ifdef GNNB_ZF_ABLATE
    if (dbg & 64)
        return; // (launch + tables + weights + first DMA landed)
#endif
    phase_p0(cur, 0, tid, 0);
    // (the W1 slice, first needed by M1: its loads had the whole prologue to land; pinned HERE so that no wait for it
    // is left inside the stage loop -- see the note on w0r above)
#pragma unroll
    for (int q = 0; q < KQ1 * 4; q++)
        asm volatile("" : "+v"(w1r[q]));
    g2_barrier();
    auto take_plan = [&]() {
        ZfStage st;
        const int4 q0 = SPLAN[0], q1 = SPLAN[1];
        st.ok = __builtin_amdgcn_readfirstlane(q0.x);
        st.chunk = __builtin_amdgcn_readfirstlane(q0.y);
        st.nb = __builtin_amdgcn_readfirstlane(q0.z);
        st.rows = __builtin_amdgcn_readfirstlane(q0.w);
        st.ga = __builtin_amdgcn_readfirstlane(q1.x);
        st.gb = __builtin_amdgcn_readfirstlane(q1.y);
        st.e0 = __builtin_amdgcn_readfirstlane(q1.z);
        st.ne = __builtin_amdgcn_readfirstlane(q1.w);
        return st;
    };
    ZfStage nxt = take_plan();
    ZF_PT(0);

    // Wave priority.  Two workgroups share a CU and fp32 MFMA and VALU instructions share one issue port: a wave in a
    // narrow phase (DMA issue, P0, P1: a few hundred VALU / LDS instructions on the workgroup's critical path) that
    // competes at equal priority with the other workgroup's two MFMA-streaming waves on its SIMD gets one instruction
    // in per 32-cycle MFMA or two (measured: P1 12.8 k cycles per stage for ~400 instructions per wave).  The narrow
    // phases therefore run at raised priority and only the long M1 stream at priority 0: the matrix pipe stays fed by
    // whichever workgroup is in M1, and the other one's narrow phases cost what their instructions cost.
    __builtin_amdgcn_s_setprio(ZF_PRIO);
    int b = 0;
    const int hg0 = cur.ga; // the graphs this workgroup pools: [hg0, hg1) -- its stages' ranges are consecutive
    int hg1 = cur.gb;
    while (cur.ok) {
        // The thread index is re-made OPAQUE every stage and every per-lane quantity is derived from it again
        // (otherwise the compiler hoists dozens of loop-invariant LDS offsets out of the stage loop and spills them)
        int tv = tid;
        asm volatile("" : "+v"(tv));
        // (the wave index as a SCALAR: wave-dependent branches and DMA addresses then run on the scalar unit instead of as
        // v_cmp / exec-mask sequences and 64-bit vector address arithmetic in all sixteen waves)
        const int li = tv & 15, lg = (tv >> 4) & 3, wv = __builtin_amdgcn_readfirstlane(tv >> 6);
        const int rows = cur.rows, nb = cur.nb;
        const int units = (rows + 15) >> 4;
#ifdef GNNB_PROBE
        prows += rows;
        pgraphs += cur.gb - cur.ga;
        punits += units;
#endif
        // ---- top: the next stage's inputs start their way to LDS (ROWS: P0 of `cur` was its last reader)
#if !ZF_DMA_IN_M1
        issue_small(nxt, b ^ 1, tv & 63, wv);
        issue_rows(nxt, b ^ 1, tv & 63, wv);
#endif
        ZF_PT(1);

        // ---- M0: H = act(A0 . W0^T + b0)   (wave: column slice x row group)
        if (ZF_ON(3)) {
            const int n0c = (wv & ((1 << cs0l) - 1)) * 16 + li;
            const int rg0 = wv >> cs0l;
            const int nt0 = (ZF_K12 && KQ0 == 1 && f0 <= 12) ? 3 : 4;
            auto m0 = [&](auto nutag, int ubase) {
                constexpr int NU = decltype(nutag)::value;
                int row0[NU];
                f32x4 acc[NU];
                // (the bias is the accumulators' initial value: the lane's four consecutive columns; from LDS: four
                // registers fewer across P1)
                const float4 bias0 = *reinterpret_cast<const float4 *>(SB0 + (n0c - li) + 4 * lg);
#pragma unroll
                for (int k = 0; k < NU; k++) {
                    row0[k] = (rg0 + (ubase + k) * nrg0) * 16;
                    acc[k] = (f32x4){bias0.x, bias0.y, bias0.z, bias0.w};
                }
                zf_mma<KQ0, NU>(A0, LD0, w0r, row0, li, lg, acc, nt0);
                if (n0c < h0) { // (h0 is 32, 64 or 128: the lane's four columns are all inside when its slice is)
                    if constexpr (MX != 0) {
                        // bf16x3: the row as two bf16 planes (hi | mid), the lane's four columns = 8 B in each
#pragma unroll
                        for (int k = 0; k < NU; k++) {
                            uint2 hi, mid;
                            split2x4<MX>(make_float4(act_t<ACT>(acc[k][0]), act_t<ACT>(acc[k][1]), act_t<ACT>(acc[k][2]), act_t<ACT>(acc[k][3])), hi, mid);
                            // (columns (n0c - li) + 4 lg .. + 3 = bytes 32 slice + 8 lg of the plane: chunk 2 slice + (lg >> 1), keyed)
                            char *hrow = reinterpret_cast<char *>(H) + (row0[k] + li) * ldhb + 2 * (n0c - li) + 16 * ((lg >> 1) ^ zf_bf_key(li)) + 8 * (lg & 1);
                            *reinterpret_cast<uint2 *>(hrow) = hi;
                            *reinterpret_cast<uint2 *>(hrow + 2 * h0) = mid;
                        }
                    } else {
#pragma unroll
                    for (int k = 0; k < NU; k++)
                        *reinterpret_cast<float4 *>(H + (row0[k] + li) * ldh + (ZF_SWZ ? 4 * ((((n0c - li) >> 2) + lg) ^ (li & kmask)) : (n0c - li) + 4 * lg)) =
                            make_float4(act_t<ACT>(acc[k][0]), act_t<ACT>(acc[k][1]), act_t<ACT>(acc[k][2]), act_t<ACT>(acc[k][3]));
                    }
                }
            };
            const int nu = rg0 < units ? (units - rg0 + nrg0 - 1) >> lnrg0 : 0; // (nrg0 is a power of two)

            // (units in groups of three: register budget)
            if (nu >= 3)
                m0(IntTag<3>{}, 0);
            else if (nu == 2)
                m0(IntTag<2>{}, 0);
            else if (nu == 1)
                m0(IntTag<1>{}, 0);
            if (nu == 6)
                m0(IntTag<3>{}, 3);
            else if (nu == 5)
                m0(IntTag<2>{}, 3);
            else if (nu == 4)
                m0(IntTag<1>{}, 3);
        }
        ZF_PT(2);
        g2_barrier(); // H complete
        ZF_PT(3);
#if ZF_DMA_IN_M1
        // ---- the next stage's inputs start their way to LDS HERE (ROWS: P0 of `cur` was its last reader): at the stage top
        // the ~40 scalar / vector instructions and two or three LDS-DMA issues per wave stood in front of M0 with nothing
        // beside them (0.9-2.2 k cycles per stage); here they run beside the other waves' MFMA stream, and the data still
        // has all of M1 to land
        issue_small(nxt, b ^ 1, tv & 63, wv);
        issue_rows(nxt, b ^ 1, tv & 63, wv);
#endif

        // ---- M1: Z = H . W1^T for the wave's column slice and its units: stays in the accumulators across the barrier
        // (the ONLY phase at low priority: see the note on s_setprio at the top of the stage loop)
        __builtin_amdgcn_s_setprio(0);
        const int n1c = (wv & ((1 << cs1l) - 1)) * 16 + li;
        const int rg1 = wv >> cs1l;
        const int nu1 = rg1 < units ? (units - rg1 + nrg1 - 1) >> lnrg1 : 0;
        constexpr int ZMAX = (ZF_UNITS * 8 + NW - 1) / NW; // units one wave can own (all eight column slices in use)
        static_assert(ZMAX <= 6, "M0 / M1 handle up to two groups of three units per wave");
        f32x4 z[ZMAX];
#pragma unroll
        for (int k = 0; k < ZMAX; k++)
            z[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ZF_ON(2)) {
            auto m1 = [&](auto nutag, auto basetag) {
                constexpr int NU = decltype(nutag)::value, UB = decltype(basetag)::value;
                int row0[NU];
                f32x4 acc[NU];
#pragma unroll
                for (int k = 0; k < NU; k++) {
                    row0[k] = (rg1 + (UB + k) * nrg1) * 16;
                    acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if constexpr (MX != 0)
                    zf_mma_bf3<MX, KQ1 / 2, NU>(reinterpret_cast<const char *>(H), ldhb, 2 * h0, w1r, row0, li, lg, acc);
                else
                    zf_mma<KQ1, NU>(H, ldh, w1r, row0, li, lg, acc, 4, kmask);
#pragma unroll
                for (int k = 0; k < NU; k++)
                    if (UB + k < ZMAX)
                        z[UB + k < ZMAX ? UB + k : 0] = acc[k];
            };
            if (nu1 >= 3)
                m1(IntTag<3>{}, IntTag<0>{});
            else if (nu1 == 2)
                m1(IntTag<2>{}, IntTag<0>{});
            else if (nu1 == 1)
                m1(IntTag<1>{}, IntTag<0>{});
            if (ZMAX >= 6 && nu1 == 6)
                m1(IntTag<3>{}, IntTag<3>{});
            else if (nu1 == 5)
                m1(IntTag<2>{}, IntTag<3>{});
            else if (nu1 == 4)
                m1(IntTag<1>{}, IntTag<3>{});
        }
        __builtin_amdgcn_s_setprio(ZF_PRIO);
        ZF_PT(4);
        // own DMA of the next stage has landed (issued a whole M0 + M1 ago; nothing younger is outstanding except
        // nothing: the pooled stores of the previous stage are older and retire first)
        dma_wait_all();
        g2_barrier(); // everybody has read H; everybody's DMA is in
        ZF_PT(5);

        // (reduced-precision forms: the overflow contract -- a non-finite value of Z in a row of the stage, i.e. an H or W1
        // element beyond fp16's range, or non-finite inputs, is flagged; rows past the stage's end hold stale LDS and are not looked at)
        if constexpr (MX != 0) {
            RangeProbe rp;
#pragma unroll
            for (int k = 0; k < ZMAX; k++)
                if (k < nu1)
                    rp.see_vec<f32x4, 4>(z[k], (rg1 + k * nrg1) * 16 + li < rows);
            rp.report(err, err_host);
        }
        // ---- ZW: Z -> H in place
        if ((n1c - li) + 4 * lg < h1 && ZF_ON(4)) { // (h1 % 4 == 0: the lane's four columns are inside or outside together)
#pragma unroll
            for (int k = 0; k < ZMAX; k++)
                if (k < nu1)
                    *reinterpret_cast<float4 *>(H + ((rg1 + k * nrg1) * 16 + li) * ldh + (ZF_SWZ ? 4 * ((((n1c - li) >> 2) + lg) ^ (li & kmask)) : (n1c - li) + 4 * lg)) =
                        make_float4(z[k][0], z[k][1], z[k][2], z[k][3]);
        }
        g2_barrier(); // Z complete
        ZF_PT(6);

        // ---- P1 + pooling.  A TASK is (graph of the stage, column part): the stage's graphs x CS column parts are dealt
        // round robin to the waves, CS in {1, 2, 4} chosen so that every wave has a task when the stage has few graphs
        // (BASELINE config 2: 8 graphs x 2 parts on 16 waves).  Inside a task a lane holds one float4 chunk of a row,
        // Gl = h1 / (4 CS) lanes make a row, and the wave's S = 64 / Gl lane groups take the graph's rows round robin and
        // IN ORDER: out_i = act(sum_j c_ij Z_j + c_ii Z_i + b1)  (CSR order, self last, as the reference's gcn_conv),
        // summed / maxed per lane, combined across the lane groups with row-swap / DPP steps (fixed order) and stored
        // with 16-B stores (reference global_add/mean/max_pool, gnn_builder_lib.h:2709-2803).  Splitting COLUMNS, not
        // rows, between waves keeps every pooled value inside one wave: no partial results cross waves.
        const int ngr = cur.gb - cur.ga;
        int csl = 0; // log2(CS)
        {
            const int nv = h1p >> 2; // float4 chunks per row
            const bool pow2 = (nv & (nv - 1)) == 0;
            // (column parts only while the tasks fill at most HALF of the waves: the phase is bound by the instructions the
            // SIMDs have to issue -- shared with the MFMA stream of the co-resident workgroup --, not by the longest wave,
            // and every task pays ~150 instructions of set-up, combine and stores: one part per graph for the four or
            // five graphs of a BASELINE config 2 stage, 42.4 instead of 43.3 us)
            // (round 4, wide shape: letting the parts fill ALL sixteen waves -- eight graphs x two parts -- is 0.25 us SLOWER,
            // 40.15 vs 39.9 us: the row walk of a task shortens from 4.1 k to 2.9 k cycles, but every task pays its ~1.9 k
            // cycles of set-up, combine and stores, and the next stage's P0 loses its idle waves)
#ifndef ZF_P1_FILL // (development A/B: 1 = column parts until the tasks fill ALL waves)
#define ZF_P1_FILL 0
#endif
            while (pow2 && csl < 2 && (ngr << (csl + 1)) <= (ZF_P1_FILL ? G2_NW : G2_NW / 2) && (nv >> (csl + 1)) >= 4)
                csl++;
        }
        if (ZF_ON(0)) {
            typedef Vf<4> V;
            int glog2 = 2; // lanes per row: the next power of two >= chunks per part
            while ((4 << glog2) < (h1p >> csl) && glog2 < 5)
                glog2++;
            const int Gl = 1 << glog2, S = 64 >> glog2;
            // lane -> (row slot sr, chunk gl).  ds_read_b128 is served in four passes of sixteen lanes, {0-3, 12-15, 20-27},
            // {4-11, 16-19, 28-31} and the same + 32: with the plain mapping (lanes 0-15 = slot 0 ...) at sixteen lanes per
            // row every pass mixes chunks of TWO rows, whose bank windows (16 x 16 B each, rows 528 B apart) overlap unless
            // the rows are a multiple of 16 apart -- a third of the kernel's LDS cycles were bank conflicts.  For Gl = 16 a
            // row slot is therefore ONE hardware pass group (its sixteen lanes read 256 contiguous bytes: conflict-free
            // whatever the rows), the chunk is the lane's rank inside the group; the slots of one chunk are then the lanes
            // l, l ^ 4, l + 32, (l ^ 4) + 32.
            int gl = tv & (Gl - 1), sr = (tv & 63) >> glog2;
            if (Gl == 16) {
                const int l5 = tv & 31;
                const bool g1 = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
                sr = ((tv & 63) >> 5) * 2 + (g1 ? 1 : 0);
                gl = g1 ? (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16)) : (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12));
            }
            const int4 *REC = reinterpret_cast<const int4 *>(RECb + (size_t)b * rec_b);
            const char *sbase = smem + rows_b + (size_t)b * small_b;
            const float *sdinv = reinterpret_cast<const float *>(sbase);
            const int32_t *sgp = reinterpret_cast<const int32_t *>(sbase + ZF_CAP * 4);
            const int32_t *scol = reinterpret_cast<const int32_t *>(SCOLb + (size_t)b * ECAP * 4);
            const bool col_lds = cur.ne <= ECAP;
            const int e0 = cur.e0;
            const int wpart = (h1p >> csl); // columns per part
            auto reduce_graph = [&](int gi, int cpart, int r0g, int r1g) { // wave-uniform row range of graph ga + gi
                r0g = max(__builtin_amdgcn_readfirstlane(r0g) - nb, 0);
                r1g = min(__builtin_amdgcn_readfirstlane(r1g) - nb, rows);
                const bool lane_on = gl * 4 < wpart;
                const int col0 = lane_on ? cpart * wpart + gl * 4 : 0; // this lane's first column
                const char *Hl = reinterpret_cast<const char *>(H) + col0 * 4; // its chunk of row 0 (unswizzled layout)
                [[maybe_unused]] const int cx = col0 * 4; // its chunk as a byte offset inside a row: XORed into a row's swizzled base
                // (LDS byte address of this lane's chunk of row i)
                auto hrow = [&](int i) { return ZF_SWZ ? smem + ((hoff + i * ldhb + hswz(i)) ^ cx) : Hl + i * ldhb; };
                const float4 bias = *reinterpret_cast<const float4 *>(SB1 + col0);
                V sum = V::splat(0.0f), mx = V::splat(-INFINITY);
#ifdef GNNB_ZF_ABLATE
                if (dbg & (1 << 20)) // (what would P1 cost if two waves shared a graph's rows?  half of the rows: WRONG results)
                    r1g = r0g + ((r1g - r0g + 1) >> 1);
#endif
                const int n = max(r1g - r0g, 0);
                // Row loop, written for instruction count (in this phase every instruction of the wave is on the
                // workgroup's critical path, and VALU issue is what the phase is bound by): running pointers instead of
                // per-row address arithmetic, no software prefetch (its register rotation cost ten moves per row; the other
                // waves of the SIMD cover the two LDS round trips), full passes without predication and one predicated
                // tail pass, maxima through v_max_f32 directly (fmaxf adds a canonicalising v_max per operand).
                const char *prec = reinterpret_cast<const char *>(REC) + (r0g + sr) * 48;
#if ZF_SWZ
                int pself = r0g + sr; // (the row index: its address is formed per pass, the key changes with the row)
                const int dself = S;
#else
                const char *pself = Hl + (r0g + sr) * ldhb;
                const int dself = ldhb << (6 - glog2);
#endif
                const int drec = 48 << (6 - glog2);
                auto vmax_raw = [](float a, float b2) {
                    float r;
                    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b2));
                    return r;
                };
                // one pass = S rows (one per lane group), in three steps: the records, the rows, the arithmetic.  (Round 4:
                // TWO passes in flight -- records of both, rows of both, then the arithmetic -- buy 0.1 us, 38.6 vs 38.7,
                // for 118 instead of 95 VGPRs, which closes the 96-register wave slot of the other batches' kernels: not kept.)
                struct RowRec { int4 ja, ca, da; };
                struct RowDat { V n0, n1, n2, n3, self; };
                auto load_rec = [&](const char *pr) {
                    RowRec r;
                    r.ja = *reinterpret_cast<const int4 *>(pr);
                    r.ca = *reinterpret_cast<const int4 *>(pr + 16);
                    r.da = *reinterpret_cast<const int4 *>(pr + 32);
                    return r;
                };
#if ZF_SWZ
                auto load_rows = [&](const RowRec &r, int ps) {
                    RowDat d;
                    d.n0 = V::load(reinterpret_cast<const float *>(smem + (r.ja.x ^ cx))); // unused slots alias the row itself (coefficient 0)
                    d.n1 = V::load(reinterpret_cast<const float *>(smem + (r.ja.y ^ cx)));
                    d.n2 = V::load(reinterpret_cast<const float *>(smem + (r.ja.z ^ cx)));
                    d.n3 = V::load(reinterpret_cast<const float *>(smem + (r.ja.w ^ cx)));
                    d.self = V::load(reinterpret_cast<const float *>(hrow(ps)));
                    return d;
                };
#else
                auto load_rows = [&](const RowRec &r, const char *ps) {
                    RowDat d;
                    d.n0 = V::load(reinterpret_cast<const float *>(Hl + r.ja.x)); // unused slots alias the row itself (coefficient 0)
                    d.n1 = V::load(reinterpret_cast<const float *>(Hl + r.ja.y));
                    d.n2 = V::load(reinterpret_cast<const float *>(Hl + r.ja.z));
                    d.n3 = V::load(reinterpret_cast<const float *>(Hl + r.ja.w));
                    d.self = V::load(reinterpret_cast<const float *>(ps));
                    return d;
                };
#endif
                auto finish_row = [&](const RowRec &r, const RowDat &d, bool active) {
                    const int4 ca = r.ca, da = r.da;
                    V acc;
                    acc.v = bias;
                    acc = vadd(acc, vmul(d.n0, V::splat(__int_as_float(ca.x))));
                    acc = vadd(acc, vmul(d.n1, V::splat(__int_as_float(ca.y))));
                    acc = vadd(acc, vmul(d.n2, V::splat(__int_as_float(ca.z))));
                    acc = vadd(acc, vmul(d.n3, V::splat(__int_as_float(ca.w))));
                    if (da.z > 4) { // degree > 4: the rest of the CSR row (slice of `col` in LDS; two loops, see P0)
                        auto more = [&](int j) {
                            acc = vadd(acc, vmul(V::load(reinterpret_cast<const float *>(hrow(j))),
                                                 V::splat(__int_as_float(da.w) * sdinv[j])));
                        };
                        if (col_lds) {
                            for (int k = da.y + 4; k < da.y + da.z; k++)
                                more(scol[min(max(k - e0, 0), ECAP - 1)] - nb);
                        } else {
                            for (int k = da.y + 4; k < da.y + da.z; k++)
                                more(col[k] - nb);
                        }
                    }
                    acc = vadd(acc, vmul(d.self, V::splat(__int_as_float(da.x))));
                    V o;
                    o.v = make_float4(act_t<ACT>(acc.v.x), act_t<ACT>(acc.v.y), act_t<ACT>(acc.v.z), act_t<ACT>(acc.v.w));
                    if (active) {
                        sum = vadd(sum, o);
                        mx.v = make_float4(vmax_raw(mx.v.x, o.v.x), vmax_raw(mx.v.y, o.v.y), vmax_raw(mx.v.z, o.v.z), vmax_raw(mx.v.w, o.v.w));
                    }
                };
                auto one_row = [&](bool active) {
                    const RowRec r = load_rec(prec);
                    const RowDat d = load_rows(r, pself);
                    finish_row(r, d, active);
                };
#ifdef GNNB_PROBE
                const unsigned long long pl0 = clock64();
#endif
                const int nfull = n >> (6 - glog2), ntail = n & (S - 1);
                int it = 0;
#pragma unroll 1
                for (; it < nfull; it++) {
                    one_row(true);
                    prec += drec;
                    pself += dself;
                }
                if (ntail) {
                    const bool active = sr < ntail;
                    if (!active) { // (inactive lane groups re-read the graph's first row)
                        prec = reinterpret_cast<const char *>(REC) + r0g * 48;
#if ZF_SWZ
                        pself = r0g;
#else
                        pself = Hl + r0g * ldhb;
#endif
                    }
                    one_row(active);
                }
#ifdef GNNB_PROBE
                pt[10] += clock64() - pl0; // (the row loop alone)
#endif
                // combine the lane groups (same chunk, different rows): lanes l and l ^ m for the row-group bits m
                // (DPP row rotations inside a 16-lane row, then the gfx950 row swaps: no LDS round trip); one branch on
                // the group size per STEP, the eight values (four sums, four maxima) inside it
                {
                    float v[8] = {sum.v.x, sum.v.y, sum.v.z, sum.v.w, mx.v.x, mx.v.y, mx.v.z, mx.v.w};
                    auto comb = [&](int i, float o) { v[i] = i < 4 ? v[i] + o : vmax_raw(v[i], o); };
                    if (Gl <= 4) {
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            comb(i, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[i]), 0x124, 0xf, 0xf, false))); // row_ror:4
                    }
                    if (Gl <= 8) {
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            comb(i, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[i]), 0x128, 0xf, 0xf, false))); // row_ror:8
                    }
                    if (Gl == 16) { // (slots of a chunk: lanes l and l ^ 4 -- see the lane mapping above)
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            comb(i, __uint_as_float(__builtin_amdgcn_ds_swizzle(__float_as_uint(v[i]), 0x101F))); // xor 4 (bit mode: and 0x1f, xor 4)
                    } else if (Gl < 16) {
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
                            v[i] = i < 4 ? __uint_as_float(q[0]) + __uint_as_float(q[1]) : vmax_raw(__uint_as_float(q[0]), __uint_as_float(q[1]));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
                        v[i] = i < 4 ? __uint_as_float(q[0]) + __uint_as_float(q[1]) : vmax_raw(__uint_as_float(q[0]), __uint_as_float(q[1]));
                    }
                    sum.v = make_float4(v[0], v[1], v[2], v[3]);
                    mx.v = make_float4(v[4], v[5], v[6], v[7]);
                }
#ifndef ZF_P1_LEAN // (development A/B: 0 = the round-5 form of the pooled stores)
#define ZF_P1_LEAN 1
#endif
                if (sr == 0 && lane_on) {
#if ZF_P1_LEAN
                    // (n is wave-uniform: ONE branch for the empty graph instead of a select per value and pool, and the reciprocal
                    // as v_rcp + one Newton step -- 3 instructions -- instead of the 11 of an IEEE division: ~25 vector
                    // instructions less per task, and every task's set-up is on the phase's critical path)
                    float inv = 0.0f;
                    if (n > 0) {
                        const float nf = (float)n;
                        const float r0 = __builtin_amdgcn_rcpf(nf);
                        inv = __builtin_fmaf(__builtin_fmaf(-nf, r0, 1.0f), r0, r0);
                    } else
                        mx = V::splat(0.0f);
#pragma unroll
                    for (int kk = 0; kk < 3; kk++) {
                        if (kk >= np)
                            break;
                        V rr = sum; // (an empty graph: 0)
                        if (pools[kk] == GNNB_POOL_MEAN)
                            rr = vmul(sum, V::splat(inv));
                        else if (pools[kk] == GNNB_POOL_MAX)
                            rr = mx;
                        rr.store(pooled + ((size_t)(cur.ga + gi) * np + kk) * h1p + col0);
                    }
#else
#pragma unroll
                    for (int kk = 0; kk < 3; kk++) {
                        if (kk >= np)
                            break;
                        V rr = sum;
                        if (pools[kk] == GNNB_POOL_MEAN)
                            rr = n > 0 ? vmul(sum, V::splat(1.0f / (float)n)) : V::splat(0.0f);
                        else if (pools[kk] == GNNB_POOL_MAX)
                            rr = n > 0 ? mx : V::splat(0.0f);
                        rr.store(pooled + ((size_t)(cur.ga + gi) * np + kk) * h1p + col0);
                    }
#endif
                }
            };
            // two loops, not one with a choice inside: a select between the LDS table and global memory is
            // if-converted into flat loads (+ a full vmcnt/lgkmcnt drain per graph)
            const int nlds = min(ngr, GMAX);
            const int ntask = nlds << csl;
            for (int t = wv; t < ntask; t += G2_NW)
                reduce_graph(t >> csl, t & ((1 << csl) - 1), sgp[t >> csl], sgp[(t >> csl) + 1]);
            for (int gi = nlds + ((wv - nlds) & (G2_NW - 1)); gi < ngr; gi += G2_NW) // a pile of empty graphs (then csl = 0)
                reduce_graph(gi, 0, node_ptr[cur.ga + gi], node_ptr[cur.ga + gi + 1]);
        }
        ZF_PT(7);

#ifdef GNNB_ZF_ABLATE
        // development: the waves WITHOUT a graph (8 .. 15 at BASELINE config 2) run a whole M1's worth of MFMAs beside P1 (one
        // column slice x ALL units each; results dropped) -- does the row walk overlap with a matrix stream on the same SIMDs?
        if ((dbg & (1 << 21)) && wv >= G2_NW / 2) {
            __builtin_amdgcn_s_setprio(0);
            for (int u = 0; u < units; u += 3) {
                int row0[3];
                f32x4 acc[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    row0[k] = min(u + k, units - 1) * 16;
                    acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if constexpr (MX == 0)
                    zf_mma<KQ1, 3>(H, ldh, w1r, row0, li, lg, acc, 4, kmask);
#pragma unroll
                for (int k = 0; k < 3; k++)
                    asm volatile("" ::"v"(acc[k]));
            }
            __builtin_amdgcn_s_setprio(ZF_PRIO);
        }
#endif
        // ---- P0 of the NEXT stage (its rows landed before the last barrier but one), starting on the first wave that
        // had no graph to reduce
        if (nxt.ok && ZF_ON(1))
            phase_p0(nxt, b ^ 1, tv, (ngr << csl) & (G2_NW - 1));
        ZF_PT(8);
        // the stage after next: planned by ONE wave (executed by all sixteen the plan was a tenth of the kernel's vector
        // instructions), handed over through LDS
        if (wv == G2_NW - 1)
            publish(plan(nxt.chunk, STAB[tv & 63], STAB[64 + (tv & 63)], STAB[128 + (tv & 63)], tv & 63), tv & 63);
        hg1 = max(hg1, cur.gb);
        cur = nxt;
        b ^= 1;
        g2_barrier(); // A0 / REC of the next stage complete; everybody is done with Z
        nxt = take_plan();
        ZF_PT(9);
#ifdef GNNB_PROBE
        nst++;
#endif
    }
    // ---- the MLP head on the graphs this workgroup pooled (round 5: reference compute_mlp_head inside the same top as
    // compute_gnn_head and compute_global_graph_pooling, templates/model.cpp.jinja:454-530, :737-765).  The pooled rows were
    // just stored by this workgroup's own P1 waves: every wave drains its stores (they are then in the XCD's L2, which this
    // CU reads through -- none of these lines can sit in its vector L1: the kernel has not read them), one barrier, then
    // groups of four waves take tiles of 16 graphs through gnnb_head.h -- weights and pooled rows as MFMA operands from L2,
    // the 16 x width activations in the (dead) H region.  ~2 us at the end of a workgroup's life instead of a third launch.
    if (HEAD && head_dev != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g2_barrier();
        const HeadArgs &head = *head_dev; // (read HERE, through the scalar cache: nothing of it lives through the stage loop)
        constexpr int NGRP = NW / 4;
        float *spart = reinterpret_cast<float *>(H);                 // [NGRP][16][ldact]: the groups' partial tiles of layer 0
        float *sact = spart + (size_t)NGRP * 16 * head_ldact;       // [2][16][ldact]
        head_tail_run<ACT, NGRP>(pooled, hg0, hg1, head, head_out, head_ldact, spart, sact, tid, [] { g2_barrier(); });
    }
#ifdef GNNB_PROBE
    if (lane == 0 && blockIdx.x * NW < 4096) {
        unsigned long long *o = g_probe + 8 * 8192 + (blockIdx.x * NW + wave) * 16; // second half: other kernels stamp the first
        o[0] = pw0;
        o[1] = wall_clock64();
        for (int i = 0; i < 11; i++)
            o[2 + i] = pt[i];
        o[13] = clock64() - pt0;
        o[14] = (unsigned long long)nst;
        o[15] = prows | (pgraphs << 32) | (punits << 48);
    }
#endif
}

// (input widths above 16 take two MFMA k blocks per row of A0: with 176-row stages the carve would pass 160 KB, so those
// models run the 96-row shape)
// zf_shape 2 (default): the 176-row shape wherever it exists (input widths up to 16).  Until the planner and the
// aggregation tasks were trimmed the two-workgroup shape was ~1-2 us faster per launch and the choice went by
// the promised graph size; since then the wide shape is the faster one alone (40.0 vs 41.3 us at BASELINE config 2) and
// in the three-stream pipeline (47-49 vs 52 us per step: its 143 KB leave room for the small kernels of the other batches).
static bool zf_wide_shape(int f0, int promise)
{
    const int sh = options().zf_shape;
    (void)promise;
    return f0 <= 16 && sh != 0;
}
static constexpr int ZF_TCAP = 62; // tiles per workgroup: the run's table lives in one register per lane (+ its end)
#endif // GNNB_ZF_KERNEL_DEFINED

#if !ZF_TU_HEAD
int zf_stage_rows(int f0, int promise) { return zf_wide_shape(f0, promise) ? 176 : 96; }
long gcn2_zf_tile_capacity(int f0, int promise)
{
    return (long)ZF_TCAP * (zf_wide_shape(f0, promise) ? 1 : 2) * device_cu_count();
}

#ifdef GNNB_ZF_ABLATE
// development only: a ring of 256 {min start, max end} slots, one per launch; gnnb_zf_dbg_spans copies them out
static unsigned long long *g_zf_spans = nullptr;
static int g_zf_launches = 0;
static unsigned long long *zf_dbg_span_slot()
{
    if (!g_zf_spans)
        return nullptr;
    return g_zf_spans + 2 * (g_zf_launches++ & 255);
}
extern "C" int gnnb_zf_dbg_reset()
{
    if (!g_zf_spans)
        (void)hipMalloc((void **)&g_zf_spans, 256 * 2 * sizeof(unsigned long long));
    unsigned long long init[512];
    for (int i = 0; i < 256; i++) {
        init[2 * i] = ~0ull;
        init[2 * i + 1] = 0ull;
    }
    g_zf_launches = 0;
    (void)hipDeviceSynchronize();
    return (int)hipMemcpy(g_zf_spans, init, sizeof(init), hipMemcpyHostToDevice);
}
extern "C" int gnnb_zf_dbg_spans(unsigned long long *host, int *launches)
{
    *launches = g_zf_launches;
    return g_zf_spans ? (int)hipMemcpy(host, g_zf_spans, 256 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) : -1;
}
#endif

#endif // !ZF_TU_HEAD

#if ZF_TU_HEAD
#define ZF_LAUNCH_NAME launch_gcn2_zf_head // (the kernels with the MLP-head tail: called by launch_gcn2_zf when the head is to run inside)
#else
#define ZF_LAUNCH_NAME launch_gcn2_zf
#endif
hipError_t ZF_LAUNCH_NAME(const BatchTables &t, const float *x, int f0, const float *w0, const float *b0,
                          int h0, const float *w1, const float *b1, int h1, int act,
                          const int32_t *pools, int num_pools, float *pooled, hipStream_t s, const float *w1f,
                          const HeadArgs *head_in, const HeadArgs *head_dev_in, float *head_out, bool *head_fused)
{
    if (head_fused)
        *head_fused = false;
    const Options &o = options();
    // (math = 1, the opt-in bf16x6 mode, does not switch this kernel off: its fp32 form is faster than the bf16x6 form of
    // k_gcn2_fused -- 39 vs 43 us at BASELINE config 2 -- and the mode must never be slower than the default)
    if (!o.fuse_gcn2 || !o.fuse_zf || t.num_nodes <= 0)
        return hipErrorNotSupported;
    const int cap = zf_stage_rows(f0, t.max_graph_nodes_hint);
    if (t.max_graph_nodes_hint <= 0 || t.max_graph_nodes_hint + t.tile_rows - 1 > cap)
        return hipErrorNotSupported; // no promise that whole graphs fit a stage
    if (f0 < 1 || f0 > 32 || !(h0 == 32 || h0 == 64 || h0 == 128) || h1 < 4 || h1 > 128 || (h1 & 3))
        return hipErrorNotSupported;
    if (w1f && (((uintptr_t)w1f) & 15))
        w1f = nullptr;
    if ((((uintptr_t)w1) & 15) || (((uintptr_t)pooled) & 15) || (((uintptr_t)x) & 3) || (b1 && (((uintptr_t)b1) & 15)))
        return hipErrorNotSupported;
    const int kq0 = f0 <= 16 ? 1 : 2, kq1 = h0 / 16;
    const int gmax = cap <= 96 ? 64 : 128;
    const int xs_b = ((cap * f0 * 4) + 15) & ~15;
    const int rows_b = xs_b + cap * 32, small_b = cap * 4 + ((gmax + 1) * 4 + 15) / 16 * 16;
#if ZF_SWZ
    const int ldh = (h0 > 64 || h1 > 64) ? 128 : ((h0 > 32 || h1 > 32) ? 64 : 32); // (as the kernel's carve)
    const size_t hoff = ((size_t)rows_b + 2 * (size_t)small_b + (size_t)cap * 16 * kq0 * 4 + 511) & ~(size_t)511;
#else
    const int ldh = (h0 > h1 ? h0 : h1) + 4;
    const size_t hoff = (size_t)rows_b + 2 * (size_t)small_b + (size_t)cap * 16 * kq0 * 4;
#endif
    const int ecap = cap <= 96 ? 512 : 1024;
    const size_t lds = hoff + (size_t)cap * ldh * 4 +
                       2 * (size_t)cap * 48 + 2 * (size_t)ecap * 4 + 512 + 32 + 512 + 768;

    if (lds > 160 * 1024)
        return hipErrorNotSupported;
#ifdef GNNB_ZF_ABLATE
    // development: GNNB_ZF_ONE=1 asks for > 80 KB of LDS so that only ONE 8-wave workgroup fits a CU (does the MFMA phase
    // of two waves per SIMD saturate the matrix pipe on its own?)
    const size_t lds_req = getenv("GNNB_ZF_ONE") && atoi(getenv("GNNB_ZF_ONE")) ? std::max(lds, (size_t)96 * 1024) : lds;
#define lds lds_req
#endif
    const int p0 = pools[0], p1 = num_pools > 1 ? pools[1] : 0, p2 = num_pools > 2 ? pools[2] : 0;
    // the MLP head inside the kernel (the caller offers it when the head's activation is the stack's): the small form's
    // shape conditions, its input = the pooled row, and its activation tiles (one per group of four waves) inside the H region
    const HeadArgs *head_dev = nullptr;
    int head_ldact = 0;
    if (head_in && head_dev_in && head_out && o.zf_head) {
        const int ld = head_small_ldact(*head_in);
        const int groups = zf_wide_shape(f0, t.max_graph_nodes_hint) && kq0 == 1 ? 4 : 2;
        if (ld > 0 && head_in->nlin >= 2 && head_in->dims[0] == num_pools * h1 && (size_t)(groups + 2) * 16 * ld * 4 <= (size_t)cap * ldh * 4) {
            head_dev = head_dev_in;
            head_ldact = ld;
        }
    }
#if ZF_TU_HEAD
    if (head_dev == nullptr)
        return hipErrorNotSupported; // (this translation unit holds the kernels with the head tail only)
#else
    if (head_dev != nullptr)
        return launch_gcn2_zf_head(t, x, f0, w0, b0, h0, w1, b1, h1, act, pools, num_pools, pooled, s, w1f, head_in, head_dev_in, head_out, head_fused);
#endif
    hipError_t rc = hipErrorNotSupported;
    auto go3 = [&](auto atag, auto q0tag, auto q1tag, auto nwtag, auto utag, auto mxtag, auto fulltag) {
        constexpr int ACT = decltype(atag)::value, KQ0 = decltype(q0tag)::value, KQ1 = decltype(q1tag)::value;
        constexpr int NW = decltype(nwtag)::value, NU = decltype(utag)::value, MX = decltype(mxtag)::value;
        auto kern = k_gcn2_zf<ACT, KQ0, KQ1, NW, NU, MX, decltype(fulltag)::value != 0, ZF_TU_HEAD != 0>;
        if (ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds) != hipSuccess) {
            rc = hipErrorNotSupported;
            return;
        }
        // occupancy of this instantiation at this LDS size, per device
        static std::mutex mu;
        static std::map<std::pair<int, size_t>, std::pair<int, int>> occ; // (device, lds) -> (blocks per CU, CUs)
        int devid = 0;
        (void)hipGetDevice(&devid);
        int blocks = 0, cus = 256;
        {
            std::lock_guard<std::mutex> lock(mu);
            auto it = occ.find(std::make_pair(devid, lds));
            if (it == occ.end()) {
                int nb = 0;
                hipDeviceProp_t prop;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * 64, lds) != hipSuccess || nb < 1)
                    nb = 1;
                if (hipGetDeviceProperties(&prop, devid) == hipSuccess)
                    cus = prop.multiProcessorCount;
                const int want = NW == 16 ? 1 : 2;
                it = occ.emplace(std::make_pair(devid, lds), std::make_pair(nb > want ? want : nb, cus)).first;
            }
            blocks = it->second.first;
            cus = it->second.second;
        }
        long long grid = (long long)cus * blocks;
        if (grid > t.num_tiles)
            grid = t.num_tiles;
        // a workgroup keeps its run of the tile table in one register per lane: at most ZF_TCAP tiles per workgroup
        const long long min_grid = ((long long)t.num_tiles + ZF_TCAP - 1) / ZF_TCAP;
        if (grid < min_grid) {
            rc = hipErrorNotSupported;
            return;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, s, x, f0, t.node_rec, t.col, t.dinv,
                           t.tile_first, t.tile_graph, t.tile_edge, t.graph_ptr, t.num_tiles, t.num_graphs, t.num_nodes, t.num_edges, w0, b0, h0, w1, w1f, b1, h1,
                           p0, p1, p2, num_pools, pooled, head_dev, head_out, head_ldact, t.err, t.err_host_dev
#ifdef GNNB_ZF_ABLATE
                           , zf_dbg_span_slot(), getenv("GNNB_ZF_DBG") ? atoi(getenv("GNNB_ZF_DBG")) : 0
#endif
        );
        rc = hipGetLastError();
    };
    // (the wide shape also exists with the last layer's width as a literal: h1 == h0 -- see the kernel's note on widths)
    auto go2 = [&](auto atag, auto q0tag, auto q1tag, auto nwtag, auto utag, auto mxtag) {
        if constexpr (decltype(nwtag)::value == 16) {
            if (h1 == h0) {
                go3(atag, q0tag, q1tag, nwtag, utag, mxtag, IntTag<1>{});
                return;
            }
        }
        go3(atag, q0tag, q1tag, nwtag, utag, mxtag, IntTag<0>{});
    };
    auto go = [&](auto atag, auto q0tag, auto q1tag) {
        if constexpr (decltype(q0tag)::value == 1) { // (the wide shape exists for one-block input widths only)
            if (zf_wide_shape(f0, t.max_graph_nodes_hint)) {
                if (launch_math() == 2) // (the opt-in bf16x3 / f16x3 forms of M1 exist in the wide shape only: every BASELINE GCN model)
                    go2(atag, q0tag, q1tag, IntTag<16>{}, IntTag<11>{}, IntTag<1>{});
                else if (launch_math() == 3)
                    go2(atag, q0tag, q1tag, IntTag<16>{}, IntTag<11>{}, IntTag<2>{});
                else
                    go2(atag, q0tag, q1tag, IntTag<16>{}, IntTag<11>{}, IntTag<0>{});
                return;
            }
        }
        go2(atag, q0tag, q1tag, IntTag<8>{}, IntTag<6>{}, IntTag<0>{});
    };
    auto go_q = [&](auto atag) {
        if (kq0 == 1 && kq1 == 8) go(atag, IntTag<1>{}, IntTag<8>{});
        else if (kq0 == 1 && kq1 == 4) go(atag, IntTag<1>{}, IntTag<4>{});
        else if (kq0 == 1 && kq1 == 2) go(atag, IntTag<1>{}, IntTag<2>{});
        else if (kq0 == 2 && kq1 == 8) go(atag, IntTag<2>{}, IntTag<8>{});
        else if (kq0 == 2 && kq1 == 4) go(atag, IntTag<2>{}, IntTag<4>{});
        else go(atag, IntTag<2>{}, IntTag<2>{});
    };
#ifdef GNNB_DEV_FAST // development builds: only the BASELINE config 2 instantiation
    if (act == GNNB_ACT_RELU && kq0 == 1 && kq1 == 8)
        go(IntTag<GNNB_ACT_RELU>{}, IntTag<1>{}, IntTag<8>{});
#else
    GNNB_DISPATCH_ACT(act, go_q)
#endif
    if (rc == hipSuccess && head_fused)
        *head_fused = head_dev != nullptr;
    return rc;
}
#ifdef GNNB_ZF_ABLATE
#undef lds
#endif
#undef ZF_LAUNCH_NAME

} // namespace gnnb
#undef ZF_TU_HEAD
