// k_pna_first.hip -- a PNA layer with a NARROW input (F <= 12: the first layer of the BASELINE config 4 model) as ONE kernel:
// pre-NN, four-way aggregate, degree scalers and the (lin-folded) post-NN product; nothing but x in, the layer's output out
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
//
// Reference: pna_conv (gnn_builder_lib.h:1891-2157): per edge h_ij = W_pre [x_i || x_j] + b, pna_conv_agg max | min | mean | std over j
// (:1750-1834; PyG's std: SURVEY finding 5), pna_conv_concat [x_i | A | amp A | att A] (13 F wide, :1836-1876), W_post, W_lin
// (folded into one [out, 13 F] matrix at upload: DESIGN 3.8).  Rounds 2-4 ran the first layer as four launches -- p GEMM (7 us),
// narrow aggregate (17 us), and the 13F- (147 us) or, under a degree promise, 5F-wide class GEMM (97 us at 0.13 of the peak: 55
// columns are no whole chunk of anything) -- through [N, F], [N, 4F] and a row permutation in HBM.  Here, in k_conv_first's shape:
//   DMA   x rows + node records + CSR slice + the rows' degree scalers of a stage of WHOLE graphs (<= 64 rows) -> LDS, two buffers
//   PQ    per row: q_i = Wa x_i + b | p_i = Wb x_i  (2F values, eight lanes per row; W_pre lives in LDS)
//   P0    per row: h_j = q_i + p_j over its sources (CSR order), max | min | mean | std, then the 13F-wide row
//         [x | A | amp A | att A] -> A0 in MFMA fragment order
//   M     Y = act(A0 . W'^T + b'), v_mfma_f32_16x16x4_f32 with swapped operands, every wave ONE 16-column slice of W' in registers
//         (13F <= 160: ten k blocks = 40 registers), accumulators kept across one barrier and written OVER A0
//   OUT   whole rows -> HBM, 16-B non-temporal stores
// Needs the max_graph_nodes promise (whole graphs in a stage); the general 13F mathematics (no degree classes: exact for any
// degree), same statistics in the same order as k_aggregate_ring<PNA>.
#include "gnnb_stack.h"

namespace gnnb {

static constexpr int PF_NW = 8, PF_WG = PF_NW * 64, PF_CAP = 64, PF_ECAP = 512;

struct PfStage {
    int ok, nb, rows, e0, ne, next_t;
};

template <int ACT, int KQ, int CSL>
__global__ __launch_bounds__(PF_WG) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pna_first(const float *__restrict__ x, int F, const int4 *__restrict__ node_rec,
                                                        const int32_t *__restrict__ col, const float *__restrict__ amp,
                                                        const float *__restrict__ att, const int32_t *__restrict__ tile_first,
                                                        const int32_t *__restrict__ tile_edge, int num_tiles, int N, int E,
                                                        const float *__restrict__ Wpre, const float *__restrict__ bpre,
                                                        const float *__restrict__ W, int ldw, const float *__restrict__ bias, int Nout,
                                                        int glog2, float *__restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD0 = 16 * KQ + 4; // A0 row (floats), padded: conflict-free fragment reads
    const int K = 13 * F;            // [x | max min mean std | amp x 4 | att x 4]
    const int LDY = Nout + 4;        // output-tile row (floats), written OVER A0 (Nout <= 16 KQ)
    const int F2 = 2 * F, LDP = F2 + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // ---- LDS carve: two input buffers {x rows | records | CSR slice | amp | att}, W_pre + b_pre, PQ, A0 (-> the output tile)
    const int xs_b = ((PF_CAP * F * 4) + 15) & ~15;
    const int rec_o = xs_b, col_o = rec_o + PF_CAP * 32, amp_o = col_o + PF_ECAP * 4, att_o = amp_o + PF_CAP * 4, in_b = att_o + PF_CAP * 4;
    float *SW = reinterpret_cast<float *>(smem + 2 * (size_t)in_b);        // W_pre [F][2F] then b_pre [F]
    float *PQ = SW + ((F * F2 + F + 3) & ~3);                              // [CAP][LDP]: q (F) | p (F)
    float *A0 = PQ + ((PF_CAP * LDP + 3) & ~3);
    float *YT = A0;

    int t0, t1;

    run_cuts(blockIdx.x, gridDim.x, (unsigned)num_tiles, t0, t1); // (32-bit: gnnb_device.h)
    if (t1 <= t0)
        return;
    const int ti = min(t0 + min(lane, t1 - t0), num_tiles);
    const int tf = min(max(tile_first[ti], 0), N), te = min(max(tile_edge[ti], 0), E);

    auto plan = [&](int ts) {
        PfStage st;
        st.ok = ts < t1 ? 1 : 0;
        st.nb = st.rows = st.e0 = st.ne = 0;
        st.next_t = ts;
        if (!st.ok)
            return st;
        const int rel = ts - t0;
        const int nb = __builtin_amdgcn_readlane(tf, rel), e0 = __builtin_amdgcn_readlane(te, rel);
        const unsigned long long fit = __ballot(lane > rel && lane <= t1 - t0 && tf - nb <= PF_CAP);
        st.nb = nb;
        st.e0 = e0;
        int endl = rel + 1; // (nothing fits: the next tile alone, cut to the stage -- only if the max_graph_nodes promise is broken)
        if (fit) {
            const unsigned long long nofit = ~fit & (~0ull << (rel + 1));
            endl = nofit ? __builtin_ctzll(nofit) - 1 : 63 - __builtin_clzll(fit);
        }
        st.rows = min(max(__builtin_amdgcn_readlane(tf, endl) - nb, 0), PF_CAP);
        st.ne = max(__builtin_amdgcn_readlane(te, endl) - e0, 0);
        st.next_t = t0 + endl;
        return st;
    };
    int vm = 0; // vector-memory instructions this wave has issued (DMA + stores): counted waits (VM operations retire in order)
    auto issue = [&](const PfStage &st, int bb) {
        if (!st.ok || st.rows <= 0)
            return;
        char *base = smem + (size_t)bb * in_b;
        const int nx = st.rows * F;
        for (int c = wave * 64; c < nx; c += PF_NW * 64, vm++)
            if (c + lane < nx)
                dma4_to_lds_u(x + (size_t)st.nb * F + c + lane, base + (size_t)c * 4);
        const char *grec = reinterpret_cast<const char *>(node_rec + 2 * (size_t)st.nb);
        const int rbytes = st.rows * 32;
        for (int c = ((wave + 2) & (PF_NW - 1)) * 1024; c < rbytes; c += PF_NW * 1024, vm++)
            if (c + lane * 16 < rbytes)
                dma16_to_lds_u(grec + c + lane * 16, base + rec_o + c);
        if (st.ne <= PF_ECAP)
            for (int c = ((wave + 4) & (PF_NW - 1)) * 64; c < st.ne; c += PF_NW * 64, vm++)
                if (c + lane < st.ne)
                    dma4_to_lds_u(col + st.e0 + c + lane, base + col_o + (size_t)c * 4);
        if (wave == 6) { // (CAP = 64 rows: one instruction each)
            if (lane < st.rows)
                dma4_to_lds_u(amp + st.nb + lane, base + amp_o);
            vm++;
        }
        if (wave == 7) {
            if (lane < st.rows)
                dma4_to_lds_u(att + st.nb + lane, base + att_o);
            vm++;
        }
    };

    PfStage cur = plan(t0);
    issue(cur, 0);
    int mark_cur = vm;
    // W_pre [F][2F] and b_pre -> LDS (tracked loads and LDS stores: complete behind the first barrier of the stage loop)
    for (int e = tid; e < F * F2 + F; e += PF_WG)
        SW[e] = e < F * F2 ? Wpre[e] : (bpre ? bpre[e - F * F2] : 0.0f);

    // ---- wave roles: 2^CSL slices of 16 output columns (Nout = 128: eight, one per wave); wave w owns slice w mod 2^CSL for the
    // units rg, rg + NRG, ... with rg = w >> CSL.  ONE slice per wave: two (k_conv_first's shape) need 72 weight registers at
    // 13F = 143 and push the kernel past the 128 registers that let two workgroups share a CU
    constexpr int NRG = PF_NW >> CSL, UPW = (PF_CAP / 16 + NRG - 1) / NRG;
    const int cs = wave & ((1 << CSL) - 1), rg = wave >> CSL;
    // weight slice -> registers: k step t of block q multiplies stage column 16 q + lg + 4 t (A0 is stored to match)
    float wr[KQ * 4];
    {
        const int n = cs * 16 + li;
#pragma unroll
        for (int q = 0; q < KQ; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = 16 * q + lg + 4 * t;
                wr[q * 4 + t] = (n < Nout && k < K) ? W[(size_t)n * ldw + k] : 0.0f;
            }
    }
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int c0 = cs * 16 + 4 * lg;
        if (bias) {
            bq.x = c0 + 0 < Nout ? bias[c0 + 0] : 0.f;
            bq.y = c0 + 1 < Nout ? bias[c0 + 1] : 0.f;
            bq.z = c0 + 2 < Nout ? bias[c0 + 2] : 0.f;
            bq.w = c0 + 3 < Nout ? bias[c0 + 3] : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < KQ * 4; q++)
        asm volatile("" : "+v"(wr[q]));
    asm volatile("" : "+v"(bq.x), "+v"(bq.y), "+v"(bq.z), "+v"(bq.w));
    const int nsteps = (K + 3) >> 2; // k steps that hold a column
    const int G = 1 << glog2, RPI = 64 >> glog2;
    auto fpos = [](int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }; // stage column k = 16 q + lg' + 4 t' sits at 16 q + 4 lg' + t'

    int b = 0;
    while (cur.ok) {
        const int rows = cur.rows, nb = cur.nb;
        const char *base = smem + (size_t)b * in_b;
        const float *xs = reinterpret_cast<const float *>(base);
        const int4 *srec = reinterpret_cast<const int4 *>(base + rec_o);
        const int32_t *scol = reinterpret_cast<const int32_t *>(base + col_o);
        const float *samp = reinterpret_cast<const float *>(base + amp_o), *satt = reinterpret_cast<const float *>(base + att_o);
        const bool col_lds = cur.ne <= PF_ECAP;
        // ---- the stage's inputs have landed (own share; then everybody's); everybody is done with the output tile and the other buffer
        vmcnt_wait_n(min(vm - mark_cur, 63));
        g2_barrier();
        const PfStage nxt = plan(cur.next_t);
        issue(nxt, b ^ 1);
        const int mark_nxt = vm;

        // ---- PQ: q_i = Wa x_i + b (o < F), p_i = Wb x_i (o >= F); eight lanes per row, lane l8 takes outputs l8, l8 + 8, ...
        {
            const int l8 = tid & 7;
            for (int i = tid >> 3; i < rows; i += PF_WG / 8) {
                // (the row and a weight row as straight-line batches of LDS reads: as a loop over a run-time F every product waited
                // for its own two reads -- 33 dependent round trips per lane and stage)
                float xr[12];
#pragma unroll
                for (int f = 0; f < 12; f++)
                    xr[f] = xs[i * F + (f < F ? f : 0)];
                for (int o = l8; o < F2; o += 8) {
                    const int r = o < F ? o : o - F, c0 = o < F ? 0 : F;
                    const float *wrow = SW + r * F2 + c0;
                    float wv[12];
#pragma unroll
                    for (int f = 0; f < 12; f++)
                        wv[f] = wrow[f < F ? f : 0];
                    float s = o < F ? SW[F * F2 + o] : 0.0f;
#pragma unroll
                    for (int f = 0; f < 12; f++)
                        s += f < F ? wv[f] * xr[f] : 0.0f;
                    PQ[i * LDP + o] = s;
                }
            }
        }
        g2_barrier();
        // ---- P0: the four statistics of h_j = q_i + p_j over the row's sources, then [x | A | amp A | att A] -> A0
        {
            const int l8 = tid & 7;
            for (int i = tid >> 3; i < rows; i += PF_WG / 8) {
                const int4 r0 = srec[2 * i], r1 = srec[2 * i + 1];
                const int deg = r0.y;
                const int jg[4] = {r0.z, r0.w, r1.x, r1.y}; // batch-global ids; unused slots alias the row itself
                const float am = samp[i], at = satt[i];
                for (int f = l8; f < F; f += 8) {
                    const float q = PQ[i * LDP + f];
                    float vmx = 0.0f, vmn = 0.0f, s1 = 0.0f, s2 = 0.0f;
                    auto take = [&](int j, bool first) {
                        const float h = q + PQ[min(max(j - nb, 0), PF_CAP - 1) * LDP + F + f];
                        if (first) {
                            vmx = h;
                            vmn = h;
                        } else {
                            vmx = fmaxf(vmx, h);
                            vmn = fminf(vmn, h);
                        }
                        s1 += h;
                        s2 += h * h;
                    };
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (deg > k)
                            take(jg[k], k == 0);
                    if (deg > 4) { // (two loops: a select between an LDS and a global pointer becomes a flat load)
                        if (col_lds) {
                            for (int k = r0.x + 4; k < r0.x + deg; k++)
                                take(scol[min(max(k - cur.e0, 0), PF_ECAP - 1)], false);
                        } else {
                            for (int k = r0.x + 4; k < r0.x + deg; k++)
                                take(col[k], false);
                        }
                    }
                    float mean = 0.0f, sd = 0.0f;
                    if (deg > 0) {
                        const float dn = (float)deg;
                        mean = s1 / dn;
                        sd = pyg_std1(s2 / dn, mean);
                    }
                    float *a = A0 + i * LD0;
                    const float st4[4] = {vmx, vmn, mean, sd};
                    a[fpos(f)] = xs[i * F + f];
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        a[fpos(F + s * F + f)] = st4[s];
                        a[fpos(5 * F + s * F + f)] = st4[s] * am;
                        a[fpos(9 * F + s * F + f)] = st4[s] * at;
                    }
                }
                for (int k = K + l8; k < 16 * KQ; k += 8) // the padding columns of the last k block
                    A0[i * LD0 + fpos(k)] = 0.0f;
            }
        }
        g2_barrier(); // A0 complete

        // ---- M: the wave's slice x its units, accumulators kept across the barrier
        const int units = (rows + 15) >> 4;
        f32x4 acc[UPW];
#pragma unroll
        for (int uu = 0; uu < UPW; uu++)
            acc[uu] = (f32x4){bq.x, bq.y, bq.z, bq.w};
        // (units in PAIRS: two independent accumulator chains, the next k block's fragments of both requested before the current
        // block's MFMAs -- as one chain per unit with its fragment read in front of every four MFMAs the phase ran at 55 % of the
        // matrix rate: 64 of the kernel's 108 us)
#pragma unroll
        for (int uu = 0; uu < UPW; uu += 2) {
            const int uA = rg + uu * NRG, uB = rg + (uu + 1) * NRG;
            const bool onB = uu + 1 < UPW && uB < units; // (wave-uniform)
            if (uA < units) {
                const float *apA = A0 + (uA * 16 + li) * LD0 + 4 * lg;
                const float *apB = onB ? A0 + (uB * 16 + li) * LD0 + 4 * lg : apA;
                float4 a0 = *reinterpret_cast<const float4 *>(apA), a1 = *reinterpret_cast<const float4 *>(apB);
#pragma unroll
                for (int q = 0; q < KQ; q++) {
                    float4 n0 = a0, n1 = a1;
                    if (q + 1 < KQ) {
                        n0 = *reinterpret_cast<const float4 *>(apA + 16 * (q + 1));
                        n1 = *reinterpret_cast<const float4 *>(apB + 16 * (q + 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (q * 4 + t >= nsteps) // (wave-uniform: this k step holds no column)
                            break;
                        const float av0 = t == 0 ? a0.x : (t == 1 ? a0.y : (t == 2 ? a0.z : a0.w));
                        const float av1 = t == 0 ? a1.x : (t == 1 ? a1.y : (t == 2 ? a1.z : a1.w));
                        acc[uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q * 4 + t], av0, acc[uu], 0, 0, 0);
                        if (uu + 1 < UPW && onB)
                            acc[uu + 1 < UPW ? uu + 1 : uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q * 4 + t], av1, acc[uu + 1 < UPW ? uu + 1 : uu], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    a0 = n0;
                    a1 = n1;
                }
            }
        }
        g2_barrier(); // everybody has read A0
        {
            const int c0 = cs * 16 + 4 * lg;
#pragma unroll
            for (int uu = 0; uu < UPW; uu++) {
                const int u = rg + uu * NRG;
                if (u < units && c0 < Nout) // (Nout % 4 == 0: the lane's four columns are inside or outside together)
                    *reinterpret_cast<float4 *>(YT + (u * 16 + li) * LDY + c0) =
                        make_float4(act_t<ACT>(acc[uu][0]), act_t<ACT>(acc[uu][1]), act_t<ACT>(acc[uu][2]), act_t<ACT>(acc[uu][3]));
            }
        }
        g2_barrier(); // the stage's output tile is complete

        // ---- OUT: whole rows -> HBM
        {
            typedef Vf<4> V;
            const int grp = lane >> glog2, gl = lane & (G - 1);
            for (int rb = wave * RPI; rb < rows; rb += PF_NW * RPI) {
                const int i = rb + grp;
                vm += 1; // (the pass's first row exists: the store instruction has an active lane)
                if (i < rows && gl * 4 < Nout)
                    agg_store<true>(V::load(YT + i * LDY + gl * 4), Y + (size_t)(nb + i) * Nout + gl * 4);
            }
        }
        cur = nxt;
        mark_cur = mark_nxt;
        b ^= 1;
    }
}

// hipErrorNotSupported (nothing launched): the caller runs the p / q GEMMs, the narrow aggregate and the post-NN GEMM
hipError_t launch_pna_first(const BatchTables &t, const float *x, int F, const float *wpre, const float *bpre, const float *w, int ldw,
                            const float *bias, float *y, int Nout, int act, hipStream_t s)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    if (!options().pna_first || F < 1 || F > 12 || !(Nout == 128 || Nout == 64) || t.tile_lo != 0 || ldw < 13 * F || (((uintptr_t)x) & 3) ||
        (((uintptr_t)y) & 15) || !t.amp || !t.att)
        return hipErrorNotSupported;
    if (t.max_graph_nodes_hint <= 0 || t.max_graph_nodes_hint + t.tile_rows - 1 > PF_CAP)
        return hipErrorNotSupported; // whole graphs must fit a stage (validated on the device by graph prep: flag 8)
    // a batch with a large segment: the promise covers graphs [0, promise_graphs) only and graph prep validates nothing about the
    // rest -- those graphs need not fit a stage (round-5 advisor finding: they got clamped sources, unflagged): layer by layer
    if (t.promise_graphs < t.num_graphs || t.large_n >= 0)
        return hipErrorNotSupported;
    const int kq = (13 * F + 15) / 16;
    if (Nout > 16 * kq) // (the output tile is written over A0)
        return hipErrorNotSupported;
    const int xs_b = ((PF_CAP * F * 4) + 15) & ~15;
    const size_t in_b = (size_t)xs_b + PF_CAP * 32 + PF_ECAP * 4 + 2 * PF_CAP * 4;
    const size_t lds = 2 * in_b + (size_t)((F * 2 * F + F + 3) & ~3) * 4 + (size_t)((PF_CAP * (2 * F + 1) + 3) & ~3) * 4 + (size_t)PF_CAP * (16 * kq + 4) * 4;
    int glog2 = 0;
    while ((4 << glog2) < Nout)
        glog2++;
    const int cus = device_cu_count();
    long long grid = std::min<long long>(2LL * cus, t.num_tiles);
    if (grid < 1)
        grid = 1;
    if ((t.num_tiles + grid - 1) / grid > 62) // a workgroup keeps its run of the tile table in one register per lane
        grid = (t.num_tiles + 61) / 62;
    hipError_t rc = hipErrorNotSupported;
    auto go2 = [&](auto atag, auto qtag, auto ctag) {
        constexpr int ACT = decltype(atag)::value, KQ = decltype(qtag)::value, CSL = decltype(ctag)::value;
        auto kern = k_pna_first<ACT, KQ, CSL>;
        if (ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds) != hipSuccess)
            return;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PF_WG), lds, s, x, F, t.node_rec, t.col, t.amp, t.att, t.tile_first, t.tile_edge,
                           t.num_tiles, t.num_nodes, t.num_edges, wpre, bpre, w, ldw, bias, Nout, glog2, y);
        rc = hipGetLastError();
    };
    auto go = [&](auto atag, auto qtag) {
        if (Nout == 128)
            go2(atag, qtag, IntTag<3>{});
        else
            go2(atag, qtag, IntTag<2>{});
    };
    auto go_a = [&](auto atag) {
        switch (kq) {
        case 8: go(atag, IntTag<8>{}); break;   // F = 9 (ogbg-molhiv) .. 9
        case 9: go(atag, IntTag<9>{}); break;   // F = 10, 11 (QM9)
        case 10: go(atag, IntTag<10>{}); break; // F = 12
        default: break;                         // (narrower inputs: Nout > 16 kq was refused above for the usual widths)
        }
    };
    GNNB_DISPATCH_ACT(act, go_a)
    return rc;
}

} // namespace gnnb
