// k_pna.hip -- a full-width PNA layer's pre-NN product and its four-way aggregate in ONE kernel: the per-node messages
// p = x . Wb^T never go to HBM (round 5)
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
//
// Reference: pna_conv (gnn_builder_lib.h:1891-2157) -- per EDGE h_ij = W_pre [x_i || x_j] + b (`linear`, :1807), then
// pna_conv_agg (:1750-1834) max | min | mean | std over a node's messages.  Here W_pre [x_i || x_j] = Wa x_i + Wb x_j: the
// source half p_j = Wb x_j is a per-NODE product, and under a max_degree promise the destination
// half is folded into the post-NN's class weights, so a layer's aggregate is
//     A_i = [max | min | mean | std]_j (Wb x_j)        (PyG's std: SURVEY finding 5).
// Rounds 2-4 ran it as a GEMM that wrote p [N, F] (k_linear_wlds) and an aggregate kernel that read it back
// (k_aggregate_ring<PNA>): 52 + 65 us and 150 MB of p traffic per layer at BASELINE config 4.  This kernel keeps p on chip,
// in the shape of the conv-stack kernels:
//   DMA   the x rows + node records + CSR slice of a stage of WHOLE graphs (<= 64 rows) -> LDS, two buffers, one row per
//         LDS-DMA instruction so that rows are PADDED (conflict-free MFMA fragment reads)
//   M     P = X . Wb^T on v_mfma_f32_16x16x4_f32 with the operands swapped (a lane ends with four consecutive columns of a
//         row), every wave holding its 16-column slice of Wb in registers; P stays in the accumulators across one barrier
//   PW    P -> LDS, OVER X (nobody reads X any more)
//   AG    per destination row (a lane group of F / 4 lanes, float4 each): the four statistics of its sources' P rows, read
//         from LDS in CSR order, -> out [N, 4F] with non-temporal 16-B stores (whole 512-B pieces per row and statistic)
// Bound: vector + matrix issue (solo at config 4: 102 us; the product alone is 31 us at the fp32 MFMA peak, the aggregate phase
// ~200 vector instructions per pair of rows, and the two add on the shared issue port; the stores are 19 us of it: DESIGN 3.8).
// Needs the max_graph_nodes promise (a graph must fit a stage: no
// p row exists outside the chip) -- without it, and for the general form with its per-destination term, the layer keeps the
// two-kernel route.  Same statistics in the same order as k_aggregate_ring<PNA>; the product's summation order is the MFMA's.
#include "gnnb_stack.h"

namespace gnnb {

static constexpr int PA_NW = 8, PA_WG = PA_NW * 64, PA_CAP = 64, PA_ECAP = 512;

struct PaStage {
    int ok, nb, rows, e0, ne, next_t;
};

// MX = 2 ("f16x3", opt-in REDUCED precision: gnnb_set_option("math", 3)): the product on the fp16 matrix cores, hi + mid fp16
// pieces of both operands, three products per 32-wide k block (k_stack_zf.hip / gnnb_device.h).  The x rows arrive by DMA as
// fp32, and split by every wave that reads them the pieces would cost what the matrix cores save -- so the stage's rows are
// split ONCE, cooperatively and in place (every thread reads its eight values, barrier, writes 16 B into each of the two planes
// of the same padded row), in front of M: two barriers more per stage.  P still goes back over the rows in fp32.
__device__ __forceinline__ int pa_h3_key(int row) { return (((row & 15) + 4) >> 3) & 1; }
template <int KQ, int MX = 0>
__global__ __launch_bounds__(PA_WG, 2) void k_pna_pagg(const float *__restrict__ x, const int4 *__restrict__ node_rec,
                                                       const int32_t *__restrict__ col, const int32_t *__restrict__ tile_first,
                                                       const int32_t *__restrict__ tile_edge, int num_tiles, int N, int E,
                                                       const float *__restrict__ Wb, int ldw, float *__restrict__ out,
                                                       int32_t *__restrict__ err, int32_t *__restrict__ err_host) // MX != 0: GNNB_FLAG_RANGE (gnnb_device.h RangeProbe)
{
    constexpr int F = 16 * KQ, LDX = F + 4;             // padded row (floats)
    constexpr int G = F / 4;                            // lanes per row (float4 each)
    constexpr int GLOG2 = G == 32 ? 5 : (G == 16 ? 4 : 3), RPI = 64 / G; // rows per wave instruction
    constexpr int CSL = KQ == 8 ? 3 : (KQ == 4 ? 2 : 1); // log2(column slices of 16)
    constexpr int NRG = PA_NW >> CSL;                   // row groups: wave w owns slice w & (2^CSL - 1) for the units rg, rg + NRG, ...
    constexpr int NU = (PA_CAP / 16 + NRG - 1) / NRG;   // units a wave can own
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // ---- LDS carve: two buffers {x rows, padded -> P | node records | CSR slice}
    constexpr int xs_b = PA_CAP * LDX * 4, rec_o = xs_b, col_o = rec_o + PA_CAP * 32, in_b = col_o + PA_ECAP * 4;

    int t0, t1;

    run_cuts(blockIdx.x, gridDim.x, (unsigned)num_tiles, t0, t1); // (32-bit: gnnb_device.h)
    if (t1 <= t0)
        return;
    // window of the tile table in registers: lane l holds tile t0 + l (the launcher keeps runs below 64 tiles)
    // (clamped: the tables of a malformed batch may hold stale entries; a flagged batch must still stay in range)
    const int ti = min(t0 + min(lane, t1 - t0), num_tiles);
    const int tf = min(max(tile_first[ti], 0), N), te = min(max(tile_edge[ti], 0), E);

    // the longest run of whole tiles from tile `ts` that fits a stage; its CSR slice is staged when it fits (else rows of
    // degree > 4 read `col` from global memory)
    auto plan = [&](int ts) {
        PaStage st;
        st.ok = ts < t1 ? 1 : 0;
        st.nb = st.rows = st.e0 = st.ne = 0;
        st.next_t = ts;
        if (!st.ok)
            return st;
        const int rel = ts - t0;
        const int nb = __builtin_amdgcn_readlane(tf, rel), e0 = __builtin_amdgcn_readlane(te, rel);
        const unsigned long long fit = __ballot(lane > rel && lane <= t1 - t0 && tf - nb <= PA_CAP);
        st.nb = nb;
        st.e0 = e0;
        int endl = rel + 1; // (nothing fits: the next tile alone, cut to the stage -- only if the max_graph_nodes promise is broken)
        if (fit) {
            const unsigned long long nofit = ~fit & (~0ull << (rel + 1));
            endl = nofit ? __builtin_ctzll(nofit) - 1 : 63 - __builtin_clzll(fit);
        }
        st.rows = min(max(__builtin_amdgcn_readlane(tf, endl) - nb, 0), PA_CAP);
        st.ne = max(__builtin_amdgcn_readlane(te, endl) - e0, 0);
        st.next_t = t0 + endl;
        return st;
    };
    int vm = 0; // vector-memory instructions this wave has issued (DMA + stores): counted waits (VM operations retire in order)
    auto issue = [&](const PaStage &st, int bb) {
        if (!st.ok || st.rows <= 0)
            return;
        char *base = smem + (size_t)bb * in_b;
        // one row per instruction (G active lanes): the LDS side of an LDS-DMA is contiguous per instruction, and the rows are padded
        for (int r = wave; r < st.rows; r += PA_NW, vm++)
            if (lane < G)
                dma16_to_lds_u(x + (size_t)(st.nb + r) * F + lane * 4, base + (size_t)r * LDX * 4);
        const char *grec = reinterpret_cast<const char *>(node_rec + 2 * (size_t)st.nb);
        const int rbytes = st.rows * 32;
        for (int c = ((wave + 2) & (PA_NW - 1)) * 1024; c < rbytes; c += PA_NW * 1024, vm++)
            if (c + lane * 16 < rbytes)
                dma16_to_lds_u(grec + c + lane * 16, base + rec_o + c);
        if (st.ne <= PA_ECAP)
            for (int c = ((wave + 4) & (PA_NW - 1)) * 64; c < st.ne; c += PA_NW * 64, vm++)
                if (c + lane < st.ne)
                    dma4_to_lds_u(col + st.e0 + c + lane, base + col_o + (size_t)c * 4);
    };

    PaStage cur = plan(t0);
    issue(cur, 0);
    int mark_cur = vm;

    // ---- the wave's 16-column slice of Wb -> registers: k step t of block q multiplies input column 16 q + 4 lg + t
    const int cs = wave & ((1 << CSL) - 1), rg = wave >> CSL;
    float wr[KQ * 4];
    if constexpr (MX != 0) {
        static_assert(KQ % 2 == 0, "f16x3: whole 32-wide k blocks");
        // (per 32-wide k block the lane's eight k values 32 q + 8 lg .. + 7 of its weight row as {hi x 4 dwords, mid x 4 dwords})
        const float *wrow = Wb + (size_t)(cs * 16 + li) * ldw + 8 * lg;
#pragma unroll
        for (int q = 0; q < KQ / 2; q++) {
            u32x4 hh, mm;
            split2x8_f16(*reinterpret_cast<const float4 *>(wrow + 32 * q), *reinterpret_cast<const float4 *>(wrow + 32 * q + 4), hh, mm);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                wr[q * 8 + i] = __uint_as_float(hh[i]);
                wr[q * 8 + 4 + i] = __uint_as_float(mm[i]);
            }
        }
    } else {
        const float *wrow = Wb + (size_t)(cs * 16 + li) * ldw + 4 * lg;
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            const float4 v = *reinterpret_cast<const float4 *>(wrow + 16 * q);
            wr[q * 4 + 0] = v.x;
            wr[q * 4 + 1] = v.y;
            wr[q * 4 + 2] = v.z;
            wr[q * 4 + 3] = v.w;
        }
    }
    // (tracked loads: finished HERE, or their first use inside the stage loop is guarded by a full vmcnt(0) -- k_stack.hip)
#pragma unroll
    for (int q = 0; q < KQ * 4; q++)
        asm volatile("" : "+v"(wr[q]));

    int b = 0;
    while (cur.ok) {
        const int rows = cur.rows, nb = cur.nb;
        char *base = smem + (size_t)b * in_b;
        float *XP = reinterpret_cast<float *>(base);
        // ---- the stage's inputs have landed (own share; then everybody's), and everybody is done with the other buffer
        vmcnt_wait_n(min(vm - mark_cur, 63));
        g2_barrier();
        const PaStage nxt = plan(cur.next_t);
        issue(nxt, b ^ 1);
        const int mark_nxt = vm;

        // ---- M: P = X . Wb^T for the wave's slice and units, kept in the accumulators
        const int units = (rows + 15) >> 4;
        f32x4 acc[NU];
#pragma unroll
        for (int k = 0; k < NU; k++)
            acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (MX != 0) {
            // ---- the stage's rows -> hi + mid fp16 pieces, in place (rows past `rows` keep whatever they hold: their products are never stored)
            constexpr int IPR = F / 8;                                  // items of eight values per row
            constexpr int IPT = (PA_CAP * IPR + PA_WG - 1) / PA_WG;     // items per thread
            float4 v0[IPT], v1[IPT];
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                const int it = min(tid + i * PA_WG, PA_CAP * IPR - 1), row = it / IPR, c8 = it % IPR;
                v0[i] = *reinterpret_cast<const float4 *>(XP + row * LDX + c8 * 8);
                v1[i] = *reinterpret_cast<const float4 *>(XP + row * LDX + c8 * 8 + 4);
            }
            g2_barrier();
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                const int it = tid + i * PA_WG, row = it / IPR, c8 = it % IPR;
                if (it < PA_CAP * IPR && row < rows) {
                    u32x4 hh, mm;
                    split2x8_f16(v0[i], v1[i], hh, mm);
                    char *pr = reinterpret_cast<char *>(XP) + row * (LDX * 4) + ((c8 ^ pa_h3_key(row)) << 4);
                    *reinterpret_cast<u32x4 *>(pr) = hh;
                    *reinterpret_cast<u32x4 *>(pr + 2 * F) = mm;
                }
            }
            g2_barrier();
#pragma unroll
            for (int k = 0; k < NU; k++) {
                const int u = rg + k * NRG;
                if (u < units) { // (wave-uniform)
                    const char *ap = reinterpret_cast<const char *>(XP) + (u * 16 + li) * (LDX * 4) + ((lg ^ pa_h3_key(li)) << 4);
                    u32x4 ah[KQ / 2], am[KQ / 2];
#pragma unroll
                    for (int q = 0; q < KQ / 2; q++) {
                        ah[q] = *reinterpret_cast<const u32x4 *>(ap + 64 * q);
                        am[q] = *reinterpret_cast<const u32x4 *>(ap + 2 * F + 64 * q);
                    }
#pragma unroll
                    for (int q = 0; q < KQ / 2; q++) {
                        const u32x4 wh = {__float_as_uint(wr[q * 8 + 0]), __float_as_uint(wr[q * 8 + 1]), __float_as_uint(wr[q * 8 + 2]), __float_as_uint(wr[q * 8 + 3])};
                        const u32x4 wm = {__float_as_uint(wr[q * 8 + 4]), __float_as_uint(wr[q * 8 + 5]), __float_as_uint(wr[q * 8 + 6]), __float_as_uint(wr[q * 8 + 7])};
                        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(wm), as_f16x8(ah[q]), acc[k], 0, 0, 0);
                        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(wh), as_f16x8(am[q]), acc[k], 0, 0, 0);
                        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(wh), as_f16x8(ah[q]), acc[k], 0, 0, 0);
                    }
                }
            }
        } else
#pragma unroll
        for (int k = 0; k < NU; k++) {
            const int u = rg + k * NRG;
            if (u < units) { // (wave-uniform)
                const float *ap = XP + (u * 16 + li) * LDX + 4 * lg;
                float4 a4[KQ];
#pragma unroll
                for (int q = 0; q < KQ; q++)
                    a4[q] = *reinterpret_cast<const float4 *>(ap + 16 * q);
#pragma unroll
                for (int q = 0; q < KQ; q++) {
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q * 4 + 0], a4[q].x, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q * 4 + 1], a4[q].y, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q * 4 + 2], a4[q].z, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q * 4 + 3], a4[q].w, acc[k], 0, 0, 0);
                }
            }
        }
        // (the reduced form's overflow contract: a non-finite P in a row of the stage -- an x or Wb element beyond fp16's range --
        // is flagged; the atomic is one more vector-memory instruction of this wave: counted)
        if constexpr (MX != 0) {
            RangeProbe rp;
#pragma unroll
            for (int k = 0; k < NU; k++)
                if (rg + k * NRG < units)
                    rp.see_vec<f32x4, 4>(acc[k], (rg + k * NRG) * 16 + li < rows);
            if (rp.any())
                vm += rp.report(err, err_host);
        }
        g2_barrier(); // everybody has read X
        // ---- PW: P over X (lane (li, lg): columns 16 cs + 4 lg .. + 3 of row 16 u + li)
#pragma unroll
        for (int k = 0; k < NU; k++) {
            const int u = rg + k * NRG;
            if (u < units)
                *reinterpret_cast<float4 *>(XP + (u * 16 + li) * LDX + cs * 16 + 4 * lg) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
        }
        g2_barrier(); // P complete

        // ---- AG: max | min | mean | std over every row's sources, CSR order (k_aggregate_ring<PNA>'s AggAcc, no destination term)
        {
            typedef Vf<4> V;
            const int4 *srec = reinterpret_cast<const int4 *>(base + rec_o);
            const int32_t *scol = reinterpret_cast<const int32_t *>(base + col_o);
            const bool col_lds = cur.ne <= PA_ECAP;
            const int e0 = cur.e0;
            const int grp = lane >> GLOG2, gl = lane & (G - 1);
            const float *Pl = XP + gl * 4;
            for (int rb = wave * RPI; rb < rows; rb += PA_NW * RPI) { // (two passes per iteration: measured, nothing)
                const int i = rb + grp;
                const bool active = i < rows;
                const int ic = active ? i : rb; // (lane groups past the stage re-read the pass's first row; their stores are predicated)
                const int4 r0 = srec[2 * ic], r1 = srec[2 * ic + 1];
                const int deg = r0.y;
                const int jl[4] = {r0.z - nb, r0.w - nb, r1.x - nb, r1.y - nb}; // unused slots alias the row itself
                V h[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    h[q] = V::load(Pl + min(max(jl[q], 0), PA_CAP - 1) * LDX);
                V vmx = V::splat(0.0f), vmn = vmx, s1 = vmx, s2 = vmx;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (deg > q) {
                        if (q == 0) {
                            vmx = h[q];
                            vmn = h[q];
                        } else {
                            vmx = vmax(vmx, h[q]);
                            vmn = vmin(vmn, h[q]);
                        }
                        s1 = vadd(s1, h[q]);
                        s2 = vadd(s2, vmul(h[q], h[q]));
                    }
                }
                if (deg > 4) { // the rest of the CSR row (two loops: a select between an LDS and a global pointer becomes a flat load)
                    auto more = [&](int j) {
                        const V hv = V::load(Pl + min(max(j, 0), PA_CAP - 1) * LDX);
                        vmx = vmax(vmx, hv);
                        vmn = vmin(vmn, hv);
                        s1 = vadd(s1, hv);
                        s2 = vadd(s2, vmul(hv, hv));
                    };
                    if (col_lds) {
                        for (int k = r0.x + 4; k < r0.x + deg; k++)
                            more(scol[min(max(k - e0, 0), PA_ECAP - 1)] - nb);
                    } else {
                        for (int k = r0.x + 4; k < r0.x + deg; k++)
                            more(col[k] - nb);
                    }
                }
                V mean = V::splat(0.0f), sd = V::splat(0.0f);
                if (deg > 0) {
                    // (one reciprocal per row instead of eight divisions: <= 1 ulp from sum / count, as the ring form's MEAN kind;
                    // the four statistics' finalisation was a third of the phase's vector instructions)
                    const V inv = V::splat(1.0f / (float)deg);
                    mean = vmul(s1, inv);
                    // PyG's std with the hardware square root (v_sqrt_f32, 1 ulp: the IEEE sqrtf expands to a dozen instructions per
                    // component, a fifth of this phase -- which is bound by vector issue: ablations in DESIGN 3.8)
                    const V m2 = vmul(s2, inv);
                    auto sd1 = [](float q2, float m) {
                        float var = q2 - m * m;
                        var = var < 1e-5f ? 1e-5f : var;
                        // (the mask `std <= sqrt(1e-5)` is taken on the variance: a 1-ulp square root of the clamp value itself could land
                        // above the threshold and leave 3.2e-3 where PyG has 0)
                        return var <= 1e-5f ? 0.0f : __builtin_amdgcn_sqrtf(var);
                    };
                    sd.v = make_float4(sd1(m2.v.x, mean.v.x), sd1(m2.v.y, mean.v.y), sd1(m2.v.z, mean.v.z), sd1(m2.v.w, mean.v.w));
                }
                vm += 4; // (the pass's first row exists: every one of its four store instructions has an active lane)
                if (active) {
                    float *o = out + (size_t)(nb + i) * (4 * F) + gl * 4;
                    agg_store<true>(vmx, o);
                    agg_store<true>(vmn, o + F);
                    agg_store<true>(mean, o + 2 * F);
                    agg_store<true>(sd, o + 3 * F);
                }
            }
        }
        // (the 16-B stores are COUNTED in `vm` like the DMA: the wait at the top of the next stage must leave them in flight)
        cur = nxt;
        mark_cur = mark_nxt;
        b ^= 1;
    }
}

// hipErrorNotSupported (nothing launched): the caller runs the p GEMM + k_aggregate_ring<PNA>
hipError_t launch_pna_pagg(const BatchTables &t, const float *x, int F, const float *wb, int ldw, float *out, hipStream_t s)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    if (!options().pna_pagg || !(F == 128 || F == 64 || F == 32) || t.tile_lo != 0)
        return hipErrorNotSupported;
    // a batch with a large segment: the max_graph_nodes promise covers graphs [0, promise_graphs) only and graph prep validates
    // nothing about the rest -- those graphs need not fit a stage (round-5 advisor finding: they got clamped sources, unflagged)
    if (t.promise_graphs < t.num_graphs || t.large_n >= 0)
        return hipErrorNotSupported;
    // whole graphs must fit a stage (validated on the device by graph prep: flag 8)
    if (t.max_graph_nodes_hint <= 0 || t.max_graph_nodes_hint + t.tile_rows - 1 > PA_CAP)
        return hipErrorNotSupported;
    if ((((uintptr_t)x | (uintptr_t)wb | (uintptr_t)out) & 15) || (ldw & 3))
        return hipErrorNotSupported;
    const size_t lds = 2 * ((size_t)PA_CAP * (F + 4) * 4 + PA_CAP * 32 + PA_ECAP * 4);
    const int cus = device_cu_count();
    long long grid = std::min<long long>(2LL * cus, t.num_tiles);
    if (grid < 1)
        grid = 1;
    if ((t.num_tiles + grid - 1) / grid > 62) // a workgroup keeps its run of the tile table in one register per lane
        grid = (t.num_tiles + 61) / 62;
    hipError_t rc = hipErrorNotSupported;
    const bool h3 = launch_math() == 3; // (opt-in f16x3, REDUCED precision)
    auto go1 = [&](auto qtag, auto mxtag) {
        constexpr int KQ = decltype(qtag)::value;
        auto kern = k_pna_pagg<KQ, decltype(mxtag)::value>;
        if (ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds) != hipSuccess)
            return;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PA_WG), lds, s, x, t.node_rec, t.col, t.tile_first, t.tile_edge, t.num_tiles,
                           t.num_nodes, t.num_edges, wb, ldw, out, t.err, t.err_host_dev);
        rc = hipGetLastError();
    };
    auto go = [&](auto qtag) {
        if (h3)
            go1(qtag, IntTag<2>{});
        else
            go1(qtag, IntTag<0>{});
    };
    if (F == 128)
        go(IntTag<8>{});
    else if (F == 64)
        go(IntTag<4>{});
    else
        go(IntTag<2>{});
    return rc;
}

} // namespace gnnb
