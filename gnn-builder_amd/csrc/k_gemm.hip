// k_gemm.hip -- dense update on the fp32 matrix cores: k_linear, k_linear_dma, k_linear_reg, k_linear_wlds
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include <vector>

#include "gnnb_device.h"

namespace gnnb {

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

// RC (round 4): the row-class mode of k_linear_dma (see there) for the shapes that kernel does not take -- PNA's FIRST layer
// under a degree promise: [x | A] with F = 11, K = 55 --: rows of A and Y through rc.perm, the weight matrix and bias of the
// 128-row tile's class.  M = the length of the class-sorted space.
template <int NT, bool RC = false> // workgroup tile = 128 x (64*NT); wave tile = 64 x (32*NT)
__global__ __launch_bounds__(WG) void k_linear(GemmArgs g, const float *__restrict__ W, int ldw,
                                               const float *__restrict__ bias,
                                               const float *__restrict__ skip,
                                               float *__restrict__ Y, int M, int N, int act, RowClasses rc = RowClasses{})
{
    static_assert(BM == 128, "a row-class tile is one workgroup tile");
    if (RC) {
        const int cls = __builtin_amdgcn_readfirstlane(rc.tile_cls[blockIdx.x]);
        W += (size_t)cls * rc.w_stride;
        if (bias)
            bias += (size_t)cls * rc.bias_stride;
    }
    constexpr int BN = 64 * NT;
    constexpr int BROWS = BN / 32; // W-tile staging passes per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);         // [2][BM*LDS_LD]
    float *Bs = As + 2 * BM * LDS_LD;                    // [2][BN*LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    int arow[4]; // (RC) the rows this thread stages: the same four in every chunk
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int pos = m0 + r0 + 32 * p;
        arow[p] = RC ? (pos < M ? rc.perm[pos] : -1) : (pos < M ? pos : -1);
    }

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
            const int row = arow[p];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row >= 0) {
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
                    const int pos = m0 + wm * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    const int rowg = RC ? (pos < M ? rc.perm[pos] : -1) : (pos < M ? pos : -1);
                    if (rowg >= 0) {
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
// The swizzle key of a chunk row.  Round 6 (profiles/r06_c{4,5}_kernels_pmc.json: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.43-0.50 in
// this kernel): with slot = piece ^ (row & 7) a ds_read_b128 -- served in groups of SIXTEEN lanes over 64 banks (256 B), rows
// {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of the 32-row fragment -- puts rows r and r + 8 (mod 16) of one parity on the same slot:
// every fragment read a two-way conflict.  Rows are 128 B: the row's parity selects the half of the 256-B bank window, so the key
// must separate the EIGHT rows of one parity inside a group: key = (row >> 1) & 7 does (even rows of the first group: 0 1 6 7 2 3 4 5).
#ifndef DMA_KEY_SHIFT
#define DMA_KEY_SHIFT 1
#endif
__device__ __forceinline__ int dkey(int r) { return (r >> DMA_KEY_SHIFT) & 7; }
static constexpr int DBUF_B = (DM + DN) * BK * 4; // 32 KB: A chunk | W chunk
// MATH 1 (opt-in, gnnb_set_option("math", 1)): the same chunks, but each 16-wide k block is multiplied as six
// v_mfma_f32_32x32x16_bf16 products of an exact 3-way bf16 split of BOTH operands (see split3), the fragments split in
// the wave after the LDS read -- 24 MFMA of 8 passes instead of 32 of 16 per k block and accumulator quartet.
// MATH 2 (opt-in, gnnb_set_option("math", 3), REDUCED precision, round 5): hi + mid fp16 pieces of both operands, three
// v_mfma_f32_32x32x16_f16 products per k block -- ~22 significant bits per product, fp16's range (gnnb_device.h).
// POOL: the pooling epilogue as its own instantiation (as a run-time branch of the one kernel it cost the plain GEMM 6-8 %:
// 109 -> 101 TFLOP/s at the C4 shape, round 4)
// RC (round 4, PNA with a degree promise): the rows of A and Y are taken through a permutation that sorts them into DEGREE
// CLASSES, and every 128-row tile multiplies by the weight matrix of its class (RowClasses): PNA's scalers depend on the
// in-degree only, so [x | A | amp(d) A | att(d) A] . W^T = [x | A] . (W_x | W_1 + amp(d) W_2 + att(d) W_3)^T -- 5 F wide
// instead of 13 F.  M is then the length of the sorted space (whole tiles), perm[position] = row or -1 (padding: loads
// re-read row 0, nothing is stored).
template <int MATH, int MODE>
__global__ __launch_bounds__(DWG) void k_linear_dma(GemmArgs g, const float *__restrict__ W, int ldw,
                                                    const float *__restrict__ bias,
                                                    const float *__restrict__ skip, float *__restrict__ Y, int M,
                                                    int N, int act, int tiles_m, int tiles_n, int split_from, int split,
                                                    PoolEpilogue pe, StreamK sk, RowClasses rc, int bias_in_lds,
                                                    int32_t *__restrict__ err, int32_t *__restrict__ err_host) // MATH 2: GNNB_FLAG_RANGE (gnnb_device.h RangeProbe)
{
    constexpr bool POOL = MODE == 1, RC = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1; // 2 x 2 waves: 64 rows x 64 columns each (a 32-row slice: 1 x 4 waves, 32 columns each)
    const int total = g.cpre[g.nseg];
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t smem_a = (uint32_t)(uintptr_t)(lds_vptr)smem;
    // DMA lane geometry: an instruction covers 8 rows x eight 16-B pieces; LDS slot p of row r holds piece p ^ dkey(r)
    const int drow = lane >> 3;
    // (the key of local row r0 + drow, r0 a multiple of 8: with DMA_KEY_SHIFT 1 it carries bit 3 of r0 -- two lane constants)
    const uint32_t dpiece_b0 = (uint32_t)(((lane & 7) ^ dkey(drow)) << 4), dpiece_b1 = (uint32_t)(((lane & 7) ^ dkey(8 + drow)) << 4);

    // PERSISTENT over work items (grid = what is resident: two workgroups per CU); the chunk pipeline runs straight
    // across item boundaries.  TAIL SPLIT: tiles / CUs is rarely whole (PNA at C4: 1153 tiles on 256 CUs = 4.5 per CU,
    // paid as 5).  Tiles from `split_from` on -- the last, partial round -- are handed out as `split` (2 or 4) row
    // slices each, so that the round costs a half or a quarter tile.  A slice keeps the tile's MFMA order per output
    // element: 64 rows = one 32-row accumulator block per wave instead of two, 32 rows = 1 x 4 waves of 32 x 32.
    // STREAM-K (round 4, sk.q > 0): the tiles from `split_from` on are not sliced by rows -- a 32-row slice keeps the whole K
    // loop, whose chunks are then too small to cover the DMA round trip: at the C4 shape the quarter-tile round cost 80 us
    // where a quarter of a round is 57, and 153 us when 4 x 129 slices just missed the 512 resident workgroups.  Instead
    // their (tile, chunk) space is cut into equal runs of q chunks, one run per workgroup: a partial tile, whole tiles, a
    // partial tile.  A workgroup multiplies its run at full tile width; where it holds only a part of a tile's K it parks the
    // accumulators in sk.part[2 * workgroup + (0: the run's first tile, 1: its last)], and the LAST workgroup to arrive at a
    // tile (sk.cnt, one counter per SHARED tile, indexed by the first run that touches the tile -- a run begins inside one tile
    // at most, so the index is unique and < gridDim.x <= SK_MAX_WG whatever the tile count --, reset by that workgroup) adds
    // the parts up IN RUN ORDER and runs the epilogue:
    // deterministic, one summation order per shape.  The launcher puts ALL tiles into that space when there is at least one
    // whole round of them (split_from = 0: no last round is left), else the tiles of the partial round.
    // The bias through LDS (round 5, not in the row-class mode, whose bias changes with the tile): read from global memory
    // inside the epilogue it is a load the compiler tracks, and the s_waitcnt in front of its first use also waits for the
    // chunk DMA of the NEXT item that is in flight by then -- ~9.5 k cycles per tile with nothing of the workgroup on the
    // matrix pipe (tools/gemm_k_sweep.py: t = 56.8 us + 0.506 us K at 4 tiles per workgroup before).  Staged HERE, in front
    // of the first DMA issue; the epilogue reads it with ds_read (lgkmcnt).  The launcher adds the bytes when N is small enough.
    float *const sbias = reinterpret_cast<float *>(smem + (size_t)DNBUF * DBUF_B + 16);
    const bool bias_lds = !RC && bias != nullptr && bias_in_lds != 0;
    if (bias_lds)
        for (int i = tid; i < N; i += DWG)
            sbias[i] = bias[i];
    const int num_tiles = tiles_m * tiles_n;
    const bool skm = sk.q > 0;
    const int num_items = skm ? split_from : split_from + split * (num_tiles - split_from); // handed out round-robin
    const int sk_total = skm ? (num_tiles - split_from) * total : 0;
    const int sk_g0 = min((int)blockIdx.x * sk.q, sk_total), sk_g1 = min(sk_g0 + sk.q, sk_total);
    const int sk_t0 = sk_g0 / total;
    const int sk_nseg = sk_g1 > sk_g0 ? (sk_g1 - 1) / total - sk_t0 + 1 : 0;
    const int n_rr = (int)blockIdx.x < num_items ? (num_items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_work = n_rr + sk_nseg;
    if (n_work == 0)
        return;
    // work item v of this workgroup: rows [m0, m0 + mrows) x columns from n0, chunks [c0, c1); skt = the tail tile of a
    // stream-K run (-1: the item owns its whole K and stores from its accumulators)
    auto decode = [&](int v, int &m0, int &n0, int &mrows, int &c0, int &c1, int &skt) {
        if (v < n_rr) {
            const int it = (int)blockIdx.x + v * (int)gridDim.x;
            const bool part = it >= split_from;
            const int j = it - split_from;
            const int t = part ? split_from + j / split : it;
            mrows = part ? DM / split : DM;
            m0 = (t / tiles_n) * DM + (part ? (j % split) * mrows : 0);
            n0 = (t % tiles_n) * DN;
            c0 = 0, c1 = total, skt = -1;
        } else {
            const int tr = sk_t0 + (v - n_rr), t = split_from + tr;
            mrows = DM;
            m0 = (t / tiles_n) * DM;
            n0 = (t % tiles_n) * DN;
            c0 = max(sk_g0 - tr * total, 0), c1 = min(sk_g1 - tr * total, total);
            skt = (c0 == 0 && c1 == total) ? -1 : tr;
        }
    };

    // issue cursor: runs ahead of the multiply cursor, across item boundaries (its item's origin is decoded once per item:
    // the integer divisions are scalar instructions in front of every wave's next MFMA)
    int iss_v = 0, iss_c = 0, iss_c1 = 0, iss_buf = 0;
    int iss_m0 = 0, iss_n0 = 0, iss_mrows = 0;
    constexpr int DA_PER_ = DM / 8 / DNW;
    // (RC) the issue item's rows for this lane's DA_PER_ DMA instructions and its class's weight matrix
    int irow[DA_PER_];
    const float *iss_w = W;
    auto issue_rows = [&]() {
        if (!RC)
            return;
        // (the cursor as explicit scalars: behind the per-lane loads below the compiler no longer proved it uniform and moved
        // the whole address arithmetic of the chunk issue -- ~100 instructions per chunk -- from the scalar to the vector unit)
        iss_m0 = __builtin_amdgcn_readfirstlane(iss_m0);
        iss_n0 = __builtin_amdgcn_readfirstlane(iss_n0);
        iss_mrows = __builtin_amdgcn_readfirstlane(iss_mrows);
        iss_c = __builtin_amdgcn_readfirstlane(iss_c);
        iss_c1 = __builtin_amdgcn_readfirstlane(iss_c1);
        iss_v = __builtin_amdgcn_readfirstlane(iss_v);
        iss_buf = __builtin_amdgcn_readfirstlane(iss_buf);
#pragma unroll
        for (int i = 0; i < DA_PER_; i++) {
            const int pos = min(iss_m0 + (wave * DA_PER_ + i) * 8 + drow, M - 1);
            irow[i] = max(rc.perm[pos], 0);
        }
        iss_w = W + (size_t)__builtin_amdgcn_readfirstlane(rc.tile_cls[min(iss_m0 / DM, tiles_m - 1)]) * rc.w_stride; // (uniform: a scalar base for the DMA)
        // (consumed HERE: left pending, every use inside the chunk loop would be guarded by s_waitcnt vmcnt(0), which also
        // waits for the chunk DMA in flight -- see the note on the row scalers below)
#pragma unroll
        for (int i = 0; i < DA_PER_; i++)
            asm volatile("" : "+v"(irow[i]));
    };
    {
        int skt_;
        decode(0, iss_m0, iss_n0, iss_mrows, iss_c, iss_c1, skt_);
        issue_rows();
    }
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
        ic.valid = iss_v < n_work;
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
        ic.ga = RC ? ap + kk : ap + (size_t)m0c * lda + kk;
        ic.gw = (RC ? iss_w : W) + (size_t)n0c * ldw + koff + kk;
        ic.ra_max = M - 1 - m0c, ic.rw_max = N - 1 - n0c; // rows past M / N re-read the last valid row (never stored)
        ic.la = smem_a + (uint32_t)iss_buf * DBUF_B, ic.lw = ic.la + DM * BK * 4;
        ic.lda_b = (uint32_t)lda * 4, ic.ldw_b = (uint32_t)ldw * 4;
        if (RC) {
            // (the class lookup sits behind per-lane loads in the cursor's update: the compiler no longer proves the bases
            // uniform -- they are, and the DMA takes them as scalars)
            auto uni = [](const float *p) {
                const uint64_t v = (uint64_t)(uintptr_t)p;
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
                return reinterpret_cast<const float *>((uintptr_t)(((uint64_t)hi << 32) | lo));
            };
            ic.ga = uni(ic.ga);
            ic.gw = uni(ic.gw);
            ic.la = __builtin_amdgcn_readfirstlane(ic.la);
            ic.lw = __builtin_amdgcn_readfirstlane(ic.lw);
        }
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
                dma16_to_lds_s(ic.ga, (uint32_t)(RC ? irow[i < DA_PER_ ? i : 0] : min(r0 + drow, ic.ra_max)) * ic.lda_b + ((r0 & 8) ? dpiece_b1 : dpiece_b0), ic.la + (uint32_t)r0 * 128);
                vm++;
            }
        }
        if (i < DW_PER) {
            const int r0 = (wave * DW_PER + i) * 8;
            dma16_to_lds_s(ic.gw, (uint32_t)min(r0 + drow, ic.rw_max) * ic.ldw_b + ((r0 & 8) ? dpiece_b1 : dpiece_b0), ic.lw + (uint32_t)r0 * 128);
            vm++;
        }
    };
    auto issue_end = [&](const IssueCtx &ic) -> int { // returns vm after the chunk's DMA (its "mark"), -1 when there was none
        if (!ic.valid)
            return -1;
        iss_buf = iss_buf + 1 == DNBUF ? 0 : iss_buf + 1;
        if (++iss_c == iss_c1) {
            if (++iss_v < n_work) {
                int skt_;
                decode(iss_v, iss_m0, iss_n0, iss_mrows, iss_c, iss_c1, skt_);
                issue_rows();
            }
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
    auto run_item = [&](auto mtag, auto ntag, int m0, int n0, int rbase, int wcol, int c0, int c1, int skt, int skseg) {
        constexpr int MC = decltype(mtag)::value; // 32-row blocks of the wave
        constexpr int NT = decltype(ntag)::value; // 32-column blocks of the wave, from column `wcol` of the tile
        f32x16 acc[MC > 0 ? MC : 1][NT];
#pragma unroll
        for (int mi = 0; mi < (MC > 0 ? MC : 1); mi++)
#pragma unroll
            for (int ni = 0; ni < NT; ni++)
#pragma unroll
                for (int i = 0; i < 16; i++)
                    acc[mi][ni][i] = 0.0f;

        // (pooling epilogue: the graph ids of the rows of this wave's 32-row blocks -- lane r + 1 holds row r, lanes 0 / 33 the
        // rows just outside --, fetched HERE and consumed below like the scalers: inside the epilogue every block paid a
        // memory round trip for them)
        int gidv[MC > 0 ? MC : 1];
#pragma unroll
        for (int mi = 0; mi < (MC > 0 ? MC : 1); mi++) {
            gidv[mi] = -1;
            if (MC > 0 && POOL) {
                const int r = m0 + rbase + mi * 32 - 1 + lane;
                if (lane < 34 && r >= 0 && r < M)
                    gidv[mi] = pe.node_graph[r];
            }
            // (RC: the row this lane's accumulator block goes to -- the same registers, the two modes exclude each other)
            if (MC > 0 && RC) {
                const int pos = m0 + rbase + mi * 32 + li;
                gidv[mi] = pos < M ? rc.perm[pos] : -1;
            }
        }
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
#pragma unroll
            for (int mi = 0; mi < MC; mi++)
                asm volatile("" : "+v"(gidv[mi]));
        }

        for (int c = c0; c < c1; c++) {
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
                if (MATH == 2) {
                    // "f16x3" (opt-in, REDUCED precision: math 3): hi + mid fp16 pieces of both operands, three products
                    // (mid.hi, hi.mid, hi.hi) per 16-wide k block -- half the MFMAs and less than half the split work of the
                    // bf16x6 form, ~22 significant bits per product, fp16's range (gnnb_device.h)
#pragma unroll
                    for (int kb2 = 0; kb2 < BK / 16; kb2++) {
                        u32x4 ah[MC > 0 ? MC : 1], am[MC > 0 ? MC : 1], wh[NT], wm[NT];
                        const int piece = 4 * kb2 + 2 * lh;
#pragma unroll
                        for (int mi = 0; mi < MC; mi++) {
                            const int r = rbase + mi * 32 + li;
                            float4 f0 = *reinterpret_cast<const float4 *>(a + r * BK + ((piece ^ dkey(r)) << 2));
                            float4 f1 = *reinterpret_cast<const float4 *>(a + r * BK + (((piece + 1) ^ dkey(r)) << 2));
                            if (scaled) {
                                f0.x *= s[mi], f0.y *= s[mi], f0.z *= s[mi], f0.w *= s[mi];
                                f1.x *= s[mi], f1.y *= s[mi], f1.z *= s[mi], f1.w *= s[mi];
                            }
                            split2x8_f16(f0, f1, ah[mi], am[mi]);
                        }
#pragma unroll
                        for (int ni = 0; ni < NT; ni++) {
                            const int r = wcol + ni * 32 + li;
                            const float4 f0 = *reinterpret_cast<const float4 *>(b + r * BK + ((piece ^ dkey(r)) << 2));
                            const float4 f1 = *reinterpret_cast<const float4 *>(b + r * BK + (((piece + 1) ^ dkey(r)) << 2));
                            split2x8_f16(f0, f1, wh[ni], wm[ni]);
                        }
                        issue_part(ic, 2 * kb2);
                        issue_part(ic, 2 * kb2 + 1);
#define GNNB_DMA_F3(WP, AP)                                                                                        \
    _Pragma("unroll") for (int mi = 0; mi < MC; mi++) _Pragma("unroll") for (int ni = 0; ni < NT; ni++)             \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(WP[ni]), as_f16x8(AP[mi]), acc[mi][ni], 0, 0, 0);
                        GNNB_DMA_F3(wm, ah)
                        GNNB_DMA_F3(wh, am)
                        GNNB_DMA_F3(wh, ah)
#undef GNNB_DMA_F3
                    }
                } else if (MATH) {
                    // lane (li, lh) of a 32x32x16 bf16 MFMA holds k = 8 lh .. + 7 of row / column li for both operands:
                    // two 16-B pieces per fragment and k block
#pragma unroll
                    for (int kb2 = 0; kb2 < BK / 16; kb2++) {
                        u32x4 ah[MC > 0 ? MC : 1], am[MC > 0 ? MC : 1], al[MC > 0 ? MC : 1], wh[NT], wm[NT], wl[NT];
                        const int piece = 4 * kb2 + 2 * lh;
#pragma unroll
                        for (int mi = 0; mi < MC; mi++) {
                            const int r = rbase + mi * 32 + li;
                            float4 f0 = *reinterpret_cast<const float4 *>(a + r * BK + ((piece ^ dkey(r)) << 2));
                            float4 f1 = *reinterpret_cast<const float4 *>(a + r * BK + (((piece + 1) ^ dkey(r)) << 2));
                            if (scaled) {
                                f0.x *= s[mi], f0.y *= s[mi], f0.z *= s[mi], f0.w *= s[mi];
                                f1.x *= s[mi], f1.y *= s[mi], f1.z *= s[mi], f1.w *= s[mi];
                            }
                            split3x8(f0, f1, ah[mi], am[mi], al[mi]);
                        }
#pragma unroll
                        for (int ni = 0; ni < NT; ni++) {
                            const int r = wcol + ni * 32 + li;
                            const float4 f0 = *reinterpret_cast<const float4 *>(b + r * BK + ((piece ^ dkey(r)) << 2));
                            const float4 f1 = *reinterpret_cast<const float4 *>(b + r * BK + (((piece + 1) ^ dkey(r)) << 2));
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
                        fa[mi] = *reinterpret_cast<const float4 *>(a + r * BK + ((piece ^ dkey(r)) << 2));
                    }
                    if (scaled) { // the row scaler multiplies the A operand, as in the register-staged kernel
#pragma unroll
                        for (int mi = 0; mi < MC; mi++)
                            fa[mi].x *= s[mi], fa[mi].y *= s[mi], fa[mi].z *= s[mi], fa[mi].w *= s[mi];
                    }
#pragma unroll
                    for (int ni = 0; ni < NT; ni++) {
                        const int r = wcol + ni * 32 + li;
                        fb[ni] = *reinterpret_cast<const float4 *>(b + r * BK + ((piece ^ dkey(r)) << 2));
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
        // (f16x3: the reduced mode's overflow contract -- what this run of chunks gave, parked stream-K parts included, is looked at
        // before any epilogue; rows past M re-read valid rows.  The atomic, rare, is one more vector-memory instruction: counted)
        if constexpr (MATH == 2) {
            RangeProbe rp;
#pragma unroll
            for (int mi = 0; mi < MC; mi++)
#pragma unroll
                for (int ni = 0; ni < NT; ni++)
                    rp.see_vec<f32x16, 16>(acc[mi][ni]);
            if (rp.any())
                vm += rp.report(err, err_host);
        }

        // ---- pooling epilogue (the model's LAST conv layer): act(acc + bias) is pooled per graph instead of stored.
        // Per 32-row block of the wave: the block goes through an 8-KB scratch in the chunk buffer that was consumed last
        // (free until the next item's first barrier; 16-B chunks XOR-swizzled by the row: conflict-free both ways), then
        // lane c walks column c down the rows IN ORDER with a running sum / max and the rows' graph ids (lane r + 1 of
        // `gid`, read with v_readlane; lanes 0 / 33 hold the rows just outside the block).  A graph that lies inside the
        // block is finished here; a piece of a graph that continues outside goes to part[block][0 = reaches the block's
        // first row, 1 = only its last][column] for launch_pool_combine.  Rows past M carry id -1 and are dropped.
        if constexpr (POOL) {
            auto pool_epi = [&](auto tag) {
                constexpr int ACT = decltype(tag)::value;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // every wave has read its last fragments
                char *scr = smem + (size_t)((buf + DNBUF - 1) % DNBUF) * DBUF_B + (size_t)wave * 8192;
                const int colg = n0 + wcol + lane; // this lane's column in the row walk
                const bool col_ok = colg < N;
                const bool any_col = __ballot(col_ok) != 0;
#pragma unroll
                for (int mi = 0; mi < MC; mi++) {
                    const int blk0 = m0 + rbase + mi * 32;
                    const int gid = gidv[mi];
#pragma unroll
                    for (int ni = 0; ni < NT; ni++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int cw = ni * 32 + 8 * q + 4 * lh; // column inside the wave's 64
                            const int cg = n0 + wcol + cw;
                            float4 v = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
                            if (bias && vec && cg + 3 < N) {
                                const float4 bv = bias_lds ? *reinterpret_cast<const float4 *>(sbias + cg) : *reinterpret_cast<const float4 *>(bias + cg);
                                v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                            } else if (bias) {
                                v.x += cg + 0 < N ? bias[cg + 0] : 0.0f;
                                v.y += cg + 1 < N ? bias[cg + 1] : 0.0f;
                                v.z += cg + 2 < N ? bias[cg + 2] : 0.0f;
                                v.w += cg + 3 < N ? bias[cg + 3] : 0.0f;
                            }
                            v.x = act_t<ACT>(v.x), v.y = act_t<ACT>(v.y), v.z = act_t<ACT>(v.z), v.w = act_t<ACT>(v.w);
                            *reinterpret_cast<float4 *>(scr + li * 256 + ((((cw >> 2)) ^ (li & 15)) << 4)) = v;
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // own scratch writes (wave-private region: no barrier)
                    const int blk = blk0 >> 5;
                    float sum = 0.0f, mx = -INFINITY;
                    int cur = __builtin_amdgcn_readlane(gid, 1), nrows = 0;
                    bool open_start = cur >= 0 && __builtin_amdgcn_readlane(gid, 0) == cur;
                    auto flush = [&](int g, int n, bool os, bool oe) { // (g, n, os, oe: wave-uniform)
                        // the stores are COUNTED (wave-uniformly: the instruction issues when any lane has a column): left
                        // uncounted, the first counted wait of the next tile drained them -- a full write round trip per tile
                        // with every wave of the workgroup idle (found in the row-class mode: 341 -> 272 us there)
                        if (g >= 0 && g < pe.num_graphs && any_col)
                            vm += (!os && !oe) ? pe.np : 1;
                        if (g < 0 || g >= pe.num_graphs || !col_ok)
                            return;
                        if (!os && !oe) {
#pragma unroll
                            for (int kk = 0; kk < 3; kk++) {
                                if (kk >= pe.np)
                                    break;
                                float rr = sum;
                                if (pe.pools[kk] == GNNB_POOL_MEAN)
                                    rr = sum / (float)n;
                                else if (pe.pools[kk] == GNNB_POOL_MAX)
                                    rr = mx;
                                pe.pooled[((size_t)g * pe.np + kk) * N + colg] = rr;
                            }
                        } else {
                            pe.part[((size_t)blk * 2 + (os ? 0 : 1)) * N + colg] = make_float2(sum, mx);
                        }
                    };
                    const char *col_p = scr + ((lane & 3) << 2);
                    const int cch = lane >> 2;
#pragma unroll 1
                    for (int r8 = 0; r8 < 32; r8 += 8) { // eight rows per step: their LDS reads go out together
                        float v8[8];
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            v8[i] = *reinterpret_cast<const float *>(col_p + (r8 + i) * 256 + ((cch ^ ((r8 + i) & 15)) << 4));
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int id = __builtin_amdgcn_readlane(gid, r8 + i + 1);
                            if (id != cur) { // (wave-uniform)
                                flush(cur, nrows, open_start, false);
                                cur = id;
                                sum = 0.0f;
                                mx = -INFINITY;
                                nrows = 0;
                                open_start = false;
                            }
                            sum += v8[i];
                            mx = fmaxf(mx, v8[i]);
                            nrows++;
                        }
                    }
                    flush(cur, nrows, open_start, cur >= 0 && __builtin_amdgcn_readlane(gid, 33) == cur);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the walk's reads are done before the next block's writes
                }
            };
            GNNB_DISPATCH_ACT(act, pool_epi)
            return;
        }

        // ---- stream-K run: park the accumulators; the last workgroup at the tile adds the runs up and goes on to the epilogue.
        // The parts are exchanged between workgroups on DIFFERENT XCDs (one L2 each): stores and loads at agent scope (sc1:
        // write-through / read from the coherent level) and a wait for the stores, instead of a release fence -- which
        // writes back the whole L2 (buffer_wbl2) per wave: measured 655 us against 540 for the row slices at the C4 shape.
        if (MC == 2 && !POOL && skt >= 0) {
            // part layout: [accumulator register 0..63][lane] (256-B rows: every store / load instruction is one contiguous piece)
            float *mine = sk.part + ((size_t)(2 * blockIdx.x + skseg) * 4 + wave) * 4096 + lane;
#pragma unroll
            for (int mi = 0; mi < MC; mi++)
#pragma unroll
                for (int ni = 0; ni < NT; ni++)
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        __hip_atomic_store(mine + ((mi * NT + ni) * 16 + i) * 64, acc[mi][ni][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the parts are at the coherent level before the arrival is counted
            __syncthreads();
            int *flag = reinterpret_cast<int *>(smem + (size_t)DNBUF * DBUF_B);
            const int w_first = (skt * total) / sk.q, w_last = ((skt + 1) * total - 1) / sk.q;
            if (tid == 0)
                *flag = __hip_atomic_fetch_add(sk.cnt + w_first, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *flag == w_last - w_first;
            if (!last)
                return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // (buffer_inv: the loads below see the other XCDs' parts)
#pragma unroll
            for (int mi = 0; mi < MC; mi++)
#pragma unroll
                for (int ni = 0; ni < NT; ni++)
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        acc[mi][ni][i] = 0.0f;
            for (int wq = w_first; wq <= w_last; wq++) { // run order = k order
                const int sg = wq * sk.q < skt * total ? 1 : 0; // the tile is that workgroup's second segment when its run began in the tile before
                const float *theirs = sk.part + ((size_t)(2 * wq + sg) * 4 + wave) * 4096 + lane;
#pragma unroll
                for (int mi = 0; mi < MC; mi++)
#pragma unroll
                    for (int ni = 0; ni < NT; ni++)
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            acc[mi][ni][i] += theirs[((mi * NT + ni) * 16 + i) * 64];
            }
            if (tid == 0)
                sk.cnt[w_first] = 0; // (nobody else comes to this tile in this launch; the next launch finds it cleared)
        }

        // D = W_tile . A_tile^T: lane (li, lh) holds Y[row = m_base + li][col = n_base + 8 (reg >> 2) + 4 lh + (reg & 3)]
        // (RC: the bias of the tile's class -- rc.bias_stride floats apart, 0: one bias for all)
        const float *const bias_all = bias;
        [[maybe_unused]] const float *bias = RC && bias_all ? bias_all + (size_t)__builtin_amdgcn_readfirstlane(rc.tile_cls[min(m0 / DM, tiles_m - 1)]) * rc.bias_stride
                                                           : bias_all;
        auto epilogue = [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int mi = 0; mi < MC; mi++) {
                const int rowg = RC ? gidv[mi] : m0 + rbase + mi * 32 + li;
                if (RC ? rowg < 0 : rowg >= M)
                    continue;
#pragma unroll
                for (int ni = 0; ni < NT; ni++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int colg = n0 + wcol + ni * 32 + 8 * q + 4 * lh;
                        if (vec && colg + 3 < N) {
                            float4 v = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2],
                                                   acc[mi][ni][4 * q + 3]);
                            if (bias) {
                                const float4 bv = bias_lds ? *reinterpret_cast<const float4 *>(sbias + colg) : *reinterpret_cast<const float4 *>(bias + colg);
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
        if (vec && n0 + wcol + 32 * NT <= N) {
#pragma unroll
            for (int mi = 0; mi < MC; mi++)
                if (RC ? __ballot(gidv[mi] < 0) == 0 : m0 + rbase + mi * 32 + 32 <= M) // (RC: a block without padding rows)
                    vm += 4 * NT;
        }
    };

    for (int v = 0; v < n_work; v++) {
        int m0, n0, mrows, c0, c1, skt;
        decode(v, m0, n0, mrows, c0, c1, skt);
        // whole tile: 2 x 2 waves of 64 x 64; 64-row slice: 2 x 2 waves of 32 x 64; 32-row slice: 1 x 4 waves of 32 x 32 (round
        // 4 -- it had been 32 x 64 on two of the four waves)
        // N <= 64 (one column tile, no 32-row slices): the same row layouts with 32-column wave tiles
        if constexpr (!POOL) {
            if (N <= 64) {
                if (mrows == DM)
                    run_item(IntTag<2>{}, IntTag<1>{}, m0, n0, wm * 64, wn * 32, c0, c1, skt, v > n_rr ? 1 : 0);
                else
                    run_item(IntTag<1>{}, IntTag<1>{}, m0, n0, wm * 32, wn * 32, c0, c1, -1, 0);
                continue;
            }
        }
        if (mrows == DM)
            run_item(IntTag<2>{}, IntTag<2>{}, m0, n0, wm * 64, wn * 64, c0, c1, skt, v > n_rr ? 1 : 0);
        else if (mrows == DM / 2)
            run_item(IntTag<1>{}, IntTag<2>{}, m0, n0, wm * 32, wn * 64, c0, c1, -1, 0);
        else
            run_item(IntTag<1>{}, IntTag<1>{}, m0, n0, 0, wave * 32, c0, c1, -1, 0);
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
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    int u0, u1;
    run_cuts(blockIdx.x, gridDim.x, (unsigned)num_units, u0, u1); // (32-bit: gnnb_device.h)
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
    if (MATH == 0 && VEC_A && KQ >= 2 && launch_math() != 0 && K == 16 * KQ && N % 32 == 0 && ldw % 4 == 0 &&
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
    const int cap = KQ >= 8 ? (int)options().gemm_max_wg_per_cu : (KQ >= 4 ? 3 : 6);
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
    // tile t of the previous unit: sixteen 64-B pieces per instruction (rows li, columns 16 t + 4 lg).  (Round 4: the tiles
    // paired through a DPP rotation so that an instruction writes eight WHOLE 128-B lines -- what took k_conv_first from 2.9 to
    // 4.2 TB/s -- is SLOWER here, 35.5 vs 34.4 us at M = 73 763: this kernel writes at ~1 TB/s beside its MFMA stream, the
    // half lines cost nothing and the eight extra VALU operations per pair do.)
    auto store_prev = [&](int t) -> int {
        if (pm < M)
            *reinterpret_cast<float4 *>(Y + (size_t)pm * N + 16 * t + 4 * lg) = pv[t];
        return 1;
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
                    if (q * TPB + i < NT)
                        vm += store_prev(q * TPB + i);
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
    if (options().gemm_variant != 0 || launch_math() != 0 || !options().gemm_wlds)
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
    const int num_cus = device_cu_count();
    const int slot = 16 * K * 4;
    // ring depth: what fits beside W in the CU's 160 KiB of LDS, at most 4.  A unit is requested ns - 1 units before
    // it is needed, piece by piece inside the MFMA stream.  Measured (tools/bench_gemm.py): ns = 2 beats ns = 3 at the
    // BASELINE sizes (31.9 vs 35.8 us at M = 73 763): a wave has only 4-7 units, and the deeper ring's longer blocking
    // prologue costs more than its steadier stream gains.
    int ns = (int)((160 * 1024 - (size_t)N * K * 4) / ((size_t)4 * slot));
    ns = std::min(std::max(ns, 1), std::min((int)options().gemm_wlds_slots, 4));
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
    // the ring form (k_first.hip): whole graphs staged once for all output columns
    if (options().first_ring && skip == nullptr && lda == (cat > 0 ? cat : K)) {
        hipError_t he = launch_conv_first(t, agg_kind, eps, x, lda, K, w, ldw, bias, y, N, act, s, cat);
        if (he != hipErrorNotSupported)
            return he;
    }
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

// the pieces of graphs that cross 32-row blocks, added up in block (= row) order; zeros for empty graphs.  Graphs inside
// one block were finished by the GEMM's epilogue and are left alone.
__global__ __launch_bounds__(WG) void k_pool_combine(PoolEpilogue pe, int M, int N)
{
    const int g = blockIdx.x;
    const int g0 = min(max(pe.graph_ptr[g], 0), M), g1 = min(max(pe.graph_ptr[g + 1], g0), M);
    const int n = g1 - g0;
    const int b0 = g0 >> 5, b1 = (g1 - 1) >> 5;
    if (n > 0 && b0 == b1)
        return;
    for (int c = threadIdx.x; c < N; c += WG) {
        float sum = 0.0f, mx = n > 0 ? -INFINITY : 0.0f;
        for (int b = b0; n > 0 && b <= b1; b++) {
            const int slot = b == b0 ? 1 : 0; // (the graph's first block holds its head -- open at the end only --, every later block a piece that reaches the block's first row)
            const float2 p = pe.part[((size_t)b * 2 + slot) * N + c];
            sum += p.x;
            mx = fmaxf(mx, p.y);
        }
        for (int kk = 0; kk < pe.np; kk++) {
            float rr = sum;
            if (pe.pools[kk] == GNNB_POOL_MEAN)
                rr = n > 0 ? sum / (float)n : 0.0f;
            else if (pe.pools[kk] == GNNB_POOL_MAX)
                rr = mx;
            pe.pooled[((size_t)g * pe.np + kk) * N + c] = rr;
        }
    }
}

hipError_t launch_pool_combine(const PoolEpilogue &pe, int M, int N, hipStream_t s)
{
    if (pe.num_graphs <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_pool_combine, dim3(pe.num_graphs), dim3(WG), 0, s, pe, M, N);
    return hipGetLastError();
}

// Stream-K scratch: SK_PART_BYTES of parked accumulators + SK_CNT_INTS arrival counters (zero between launches).  A workspace
// owns its own (stream_k_scratch_create at gnnb_workspace_create, freed with it, handed to launch_linear): forwards of different
// workspaces -- on any streams, eager or replayed from hipGraphs -- never share one.  The standalone gnnb_linear entry has no
// workspace: it takes one scratch per (device, stream) from the map below -- launches on one stream run in order and may share
// it -- allocated at the first GEMM that wants it, and NEVER while the stream is being captured (a captured launch takes the
// row slices: a graph replayed on another stream, or beside an eager launch, must not carry the shared scratch's address).
// Tail-only runs (fewer tiles than resident workgroups) for K >= 1024 (32 chunks): measured at C4's 13F GEMM (52 chunks, 1153
// tiles) 112.5 against 106 TFLOP/s with row slices, at C5's K = 512 (16 chunks: runs of 7) 105 against 108 -- short runs are
// all pipeline prologue and fix-up.  With every tile in the space (at least one whole round of tiles) a run is tiles * chunks /
// 512 long, but nearly every tile then pays a fix-up (64 KB parked and read back): at K = 256 (8 chunks, 577 tiles) that took
// the GEMM from 74 to 54 TFLOP/s, at K = 512 it is a wash, at K = 1664 it is +1.5 % on a 1.13-round shape: K >= 1024 as well.
static constexpr int SK_MIN_Q = 6, SK_MIN_TOTAL = 32, SK_ALL_MIN_TOTAL = 32, SK_MAX_WG = 512, SK_MAX_STREAMS = 16;
static constexpr size_t SK_PART_BYTES = (size_t)2 * SK_MAX_WG * DM * DN * sizeof(float);
static constexpr int SK_CNT_INTS = 1024; // the counter of a shared tile is indexed by a workgroup: < grid <= SK_MAX_WG
static_assert(SK_MAX_WG <= SK_CNT_INTS, "one arrival counter per resident workgroup at least");
static constexpr size_t SK_GUARD_BYTES = 4096; // behind the counters: a fixed pattern nothing may touch (stream_k_guard_intact)
static constexpr size_t SK_TAIL_BYTES = (size_t)SK_CNT_INTS * sizeof(int) + SK_GUARD_BYTES;
size_t stream_k_scratch_bytes() { return SK_PART_BYTES + SK_TAIL_BYTES; }
hipError_t stream_k_scratch_init(void *base, hipStream_t s)
{
    char *p = reinterpret_cast<char *>(base);
    hipError_t e = hipMemsetAsync(p + SK_PART_BYTES, 0, (size_t)SK_CNT_INTS * sizeof(int), s);
    if (e == hipSuccess)
        e = hipMemsetAsync(p + SK_PART_BYTES + (size_t)SK_CNT_INTS * sizeof(int), 0xA5, SK_GUARD_BYTES, s);
    return e;
}
hipError_t stream_k_scratch_init_sync(void *base) // (workspace creation: no stream involved)
{
    char *p = reinterpret_cast<char *>(base);
    hipError_t e = hipMemset(p + SK_PART_BYTES, 0, (size_t)SK_CNT_INTS * sizeof(int));
    if (e == hipSuccess)
        e = hipMemset(p + SK_PART_BYTES + (size_t)SK_CNT_INTS * sizeof(int), 0xA5, SK_GUARD_BYTES);
    return e;
}
// 1 = counters all zero (no launch in flight on `s`) and the guard pattern whole, 0 = not, -1 = the read-back failed
static int stream_k_tail_ok(const StreamK &k, hipStream_t s)
{
    std::vector<unsigned char> h(SK_TAIL_BYTES);
    if (hipMemcpyAsync(h.data(), k.cnt, SK_TAIL_BYTES, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return -1;
    for (size_t i = 0; i < SK_TAIL_BYTES; i++)
        if (h[i] != (i < (size_t)SK_CNT_INTS * sizeof(int) ? 0x00 : 0xA5))
            return 0;
    return 1;
}
StreamK stream_k_scratch_at(void *base)
{
    StreamK k;
    k.part = reinterpret_cast<float *>(base);
    k.cnt = reinterpret_cast<int *>(reinterpret_cast<char *>(base) + SK_PART_BYTES);
    return k;
}
static std::mutex g_sk_mu;
static std::map<std::pair<int, hipStream_t>, StreamK> g_sk_have;
int stream_k_guard_intact(const StreamK *owned, hipStream_t s)
{
    if (owned && owned->cnt)
        return stream_k_tail_ok(*owned, s);
    StreamK k;
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(g_sk_mu);
        auto it = g_sk_have.find(std::make_pair(dev, s));
        if (it == g_sk_have.end())
            return 1; // (no scratch yet: nothing to damage)
        k = it->second;
    }
    return stream_k_tail_ok(k, s);
}
static bool stream_k_scratch(hipStream_t s, const StreamK *owned, StreamK &out)
{
    if (owned && owned->part && owned->cnt) {
        out = *owned;
        out.q = 0;
        return true;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return false;
    }
    std::map<std::pair<int, hipStream_t>, StreamK> &have = g_sk_have;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_sk_mu);
    auto it = have.find(std::make_pair(dev, s));
    if (it == have.end()) {
        if ((int)have.size() >= SK_MAX_STREAMS)
            return false;
        char *p = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&p), stream_k_scratch_bytes()) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (stream_k_scratch_init(p, s) != hipSuccess) { // (in stream order, in front of the first launch that counts)
            (void)hipGetLastError();
            (void)hipFree(p);
            return false;
        }
        it = have.emplace(std::make_pair(dev, s), stream_k_scratch_at(p)).first;
    }
    out = it->second;
    return true;
}

hipError_t launch_linear(const GemmArgs &g, const float *w, int ldw, const float *bias,
                         const float *skip, float *y, int M, int N, int act, hipStream_t s, const PoolEpilogue *pep,
                         const RowClasses *rcp, const StreamK *sk_owned)
{
    if (M <= 0 || N <= 0)
        return (pep || rcp) ? hipErrorNotSupported : hipSuccess;
    if (rcp && (pep || !rcp->perm || !rcp->tile_cls || M % DM != 0 || N <= 32))
        return hipErrorNotSupported; // (row classes: whole 128-row tiles, k_linear_dma or the generic kernel)
    const RowClasses rc = rcp ? *rcp : RowClasses{};
    const PoolEpilogue pe = pep ? *pep : PoolEpilogue{};
    if (!pep && !rcp) {
        if (linear_wlds_eligible(g, w, ldw, bias, skip, y, N))
            return launch_linear_wlds(g, w, ldw, bias, skip, y, M, N, act, s);
        if (linear_reg_eligible(g))
            return launch_linear_reg(g, w, ldw, bias, skip, y, M, N, act, s);
    } else if (pep && (skip != nullptr || linear_wlds_eligible(g, w, ldw, bias, skip, y, N) || linear_reg_eligible(g))) {
        return hipErrorNotSupported; // (the pooling epilogue exists in k_linear_dma: the large-K segmented GEMM)
    }
    const int gm = (M + BM - 1) / BM;
    // (N in 33 .. 64 -- the last layer of the reference's benchmark models, 128 -> 64 -- takes the same kernel with 32-column
    // wave tiles, when K is large enough to be worth the chunk pipeline: it ran in k_linear<1> at 0.20 of peak, a quarter of
    // the ref6 PNA step)
    const bool dma_narrow = N > 32 && N <= 64 && !pep && g.cpre[g.nseg] >= 8;
    if ((N > 64 || dma_narrow) && options().gemm_dma) {
        bool plain = (ldw % 4 == 0) && (((uintptr_t)w & 15) == 0);
        for (int sg = 0; sg < g.nseg && plain; sg++)
            plain = g.avec[sg] && g.wvec[sg] && (g.k[sg] % BK == 0) && (g.koff[sg] % 4 == 0);
        if (plain) {
            const int bias_in_lds = (bias && !rcp && N <= 2048) ? 1 : 0; // (8 KB at most beside the two 64-KB workgroups of a CU)
            const size_t lds = (size_t)DNBUF * DBUF_B + 16 + (bias_in_lds ? (((size_t)N * 4 + 15) & ~(size_t)15) : 0); // (+ the stream-K arrival flag, + the bias)
            {
                const int mode = pep ? 1 : (rcp ? 2 : 0);
                const int mv = launch_math() == 3 ? 2 : (launch_math() ? 1 : 0);
                const void *fns[3][3] = {
                    {reinterpret_cast<const void *>(k_linear_dma<0, 0>), reinterpret_cast<const void *>(k_linear_dma<0, 1>), reinterpret_cast<const void *>(k_linear_dma<0, 2>)},
                    {reinterpret_cast<const void *>(k_linear_dma<1, 0>), reinterpret_cast<const void *>(k_linear_dma<1, 1>), reinterpret_cast<const void *>(k_linear_dma<1, 2>)},
                    {reinterpret_cast<const void *>(k_linear_dma<2, 0>), reinterpret_cast<const void *>(k_linear_dma<2, 1>), reinterpret_cast<const void *>(k_linear_dma<2, 2>)}};
                const void *fn = fns[mv][mode];
                hipError_t e = ensure_dynamic_lds(fn, lds);
                if (e != hipSuccess)
                    return e;
            }
            const int num_cus = device_cu_count();
            const int tm = (M + DM - 1) / DM, tn = (N + DN - 1) / DN, tiles = tm * tn;
            // two 64-KB workgroups are resident per CU and share its matrix pipe: what has to come out even is the
            // work per CU.  The last, partial round of tiles (all of them when there are fewer tiles than CUs) goes out
            // in 2 or 4 row slices per tile when those still fit one round (see the kernel)
            const int rem = tiles % num_cus;
            int split = 1;
            StreamK sk;
            const int total = g.cpre[g.nseg], resident = DWGPC * num_cus;
            bool sk_all = false;
            if (options().gemm_tail_split == 2 && rem > 0 && !pep && tiles >= resident && total >= SK_ALL_MIN_TOTAL &&
                resident <= SK_MAX_WG && (long long)tiles * total < (1ll << 30) && stream_k_scratch(s, sk_owned, sk)) {
                // at least one whole round of tiles: EVERY tile goes into the (tile, chunk) space and every resident workgroup
                // takes one equal run of it (tiles * total / resident chunks: a partial tile, whole tiles, a partial tile) --
                // there is no last round left; a tile is shared by two workgroups at most
                sk.q = (int)(((long long)tiles * total + resident - 1) / resident);
                sk_all = true;
            } else if (options().gemm_tail_split == 2 && rem > 0 && !pep && total >= SK_MIN_TOTAL && resident <= SK_MAX_WG &&
                       (long long)rem * total < (1ll << 30) && stream_k_scratch(s, sk_owned, sk)) {
                // fewer tiles than resident workgroups or a K too short for the above: equal runs of the last round's space,
                // at least SK_MIN_Q chunks long (shorter ones are all pipeline prologue)
                sk.q = std::max((rem * total + resident - 1) / resident, SK_MIN_Q);
            } else if (options().gemm_tail_split && rem > 0 && !pep) { // (pooling epilogue: whole tiles only -- its blocks are 32-row aligned, every wave joins its barrier)
                split = (4 * rem <= resident && !dma_narrow) ? 4 : (2 * rem <= resident ? 2 : 1);
            }
            const int split_from = sk_all ? 0 : ((split > 1 || sk.q > 0) ? tiles - rem : tiles);
            const int grid = sk.q > 0 ? std::min(std::max(split_from, (int)(((long long)(tiles - split_from) * total + sk.q - 1) / sk.q)), resident)
                                      : std::min(split_from + split * (tiles - split_from), resident);
#define GNNB_DMA_LAUNCH(MATHV, MODEV)                                                                                    \
    hipLaunchKernelGGL((k_linear_dma<MATHV, MODEV>), dim3(grid), dim3(DWG), lds, s, g, w, ldw, bias, skip, y, M, N, act, tm, \
                       tn, split_from, split, pe, sk, rc, bias_in_lds, flagw.err, flagw.err_host)
            const FlagWord flagw = launch_flag_word(); // (the workspace whose forward this launch belongs to; none: stand-alone gnnb_linear)
            if (launch_math() == 3) { // (f16x3: opt-in, reduced precision)
                if (pep)
                    GNNB_DMA_LAUNCH(2, 1);
                else if (rcp)
                    GNNB_DMA_LAUNCH(2, 2);
                else
                    GNNB_DMA_LAUNCH(2, 0);
            } else if (launch_math()) {
                if (pep)
                    GNNB_DMA_LAUNCH(1, 1);
                else if (rcp)
                    GNNB_DMA_LAUNCH(1, 2);
                else
                    GNNB_DMA_LAUNCH(1, 0);
            } else {
                if (pep)
                    GNNB_DMA_LAUNCH(0, 1);
                else if (rcp)
                    GNNB_DMA_LAUNCH(0, 2);
                else
                    GNNB_DMA_LAUNCH(0, 0);
            }
#undef GNNB_DMA_LAUNCH
            return hipGetLastError();
        }
    }
    if (pep)
        return hipErrorNotSupported;
    if (rcp) { // (row classes outside the DMA kernel's shapes: the generic tiles, one workgroup per class tile)
        constexpr int NT = 2;
        const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * 64 * NT * LDS_LD) * 4;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_linear<NT, true>), lds);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL((k_linear<NT, true>), dim3(M / BM, (N + 127) / 128), dim3(WG), lds, s, g, w, ldw, bias, skip, y, M, N, act, rc);
        return hipGetLastError();
    }
    if (N > 64) {
        constexpr int NT = 2;
        const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * 64 * NT * LDS_LD) * 4;
        {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(k_linear<NT, false>), lds);
            if (e != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL((k_linear<NT, false>), dim3(gm, (N + 127) / 128), dim3(WG), lds, s, g, w, ldw,
                           bias, skip, y, M, N, act, RowClasses{});
    } else {
        constexpr int NT = 1;
        const size_t lds = (size_t)(2 * BM * LDS_LD + 2 * 64 * NT * LDS_LD) * 4;
        hipLaunchKernelGGL((k_linear<NT, false>), dim3(gm, 1), dim3(WG), lds, s, g, w, ldw, bias, skip, y, M,
                           N, act, RowClasses{});
    }
    return hipGetLastError();
}


} // namespace gnnb
