// k_first_mean.hip -- GraphSAGE's narrow first layer AND the next layer's mean aggregate in one kernel (round 5)
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
//
// Reference: two consecutive sage_conv layers of compute_gnn_head (templates/model.cpp.jinja:151-359, gnn_builder_lib.h:2161-2341):
// layer 0 out_i = act(W_l . mean_j x_j + b_l + W_r . x_i) on 36-byte rows, then layer 1's sage_conv_agg m_i = mean_j out_j on
// 1-KB rows.  Round 4 ran k_conv_first (writes out [N, d]) and k_aggregate_ring<MEAN> (reads it back, writes m [N, d]): 51 + 80-87
// us and a 214-MB re-read at BASELINE config 5.  Both layers gather over the SAME graphs, and k_conv_first already stages whole
// graphs: here the stage's output rows stay in LDS after the product, and the mean of every row's sources is taken from
// there -- out and m leave the chip once each, nothing is read back.
//   DMA   x rows + node records + CSR slice of a stage of WHOLE graphs (<= 56 rows) -> LDS, two buffers
//   P0    A0 = [mean_j x_j | x_i]  [rows, 2F <= 32]   eight lanes per row, LDS -> LDS (k_conv_first's P0, staged form)
//   M     Y = act(A0 . [W_l | W_r]^T + b): v_mfma_f32_16x16x4_f32, operands swapped, every wave two adjacent 16-column slices of
//         the fused weight in registers, -> the stage's output tile YT [rows, d] in LDS (padded rows)
//   OUT   per row (a lane group of d / 4 lanes): Y row -> HBM, then m = (sum of its sources' YT rows, CSR order) / degree -> HBM;
//         16-B non-temporal stores, whole rows
// Needs the max_graph_nodes promise (a graph must fit a stage: a source's output row exists only in LDS).  Same sums in the same
// order as k_conv_first + k_aggregate_ring<MEAN> (bit-identical outputs).
#include "gnnb_stack.h"

namespace gnnb {

static constexpr int FM_NW = 8, FM_WG = FM_NW * 64, FM_CAP = 56, FM_ECAP = 448;

struct FmStage {
    int ok, nb, rows, e0, ne, next_t;
};

template <int ACT, int KQ>
__global__ __launch_bounds__(FM_WG, 2) void k_sage_first_mean(const float *__restrict__ x, int F, const int4 *__restrict__ node_rec,
                                                              const int32_t *__restrict__ col, const int32_t *__restrict__ tile_first,
                                                              const int32_t *__restrict__ tile_edge, int num_tiles, int N, int E,
                                                              const float *__restrict__ W, int ldw, const float *__restrict__ bias, int K,
                                                              int Nout, int glog2, float *__restrict__ Y, float *__restrict__ Mo)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD0 = 16 * KQ + 4; // A0 row (floats), padded: conflict-free fragment reads
    const int LDY = Nout + 4;        // output-tile row (floats), padded
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // ---- LDS carve: two input buffers {x rows | records | CSR slice}, A0, the output tile
    const int xs_b = ((FM_CAP * F * 4) + 15) & ~15;
    const int rec_o = xs_b, col_o = rec_o + FM_CAP * 32, in_b = col_o + FM_ECAP * 4;
    float *A0 = reinterpret_cast<float *>(smem + 2 * (size_t)in_b);
    float *YT = A0 + FM_CAP * LD0;

    int t0, t1;

    run_cuts(blockIdx.x, gridDim.x, (unsigned)num_tiles, t0, t1); // (32-bit: gnnb_device.h)
    if (t1 <= t0)
        return;
    // window of the tile table in registers: lane l holds tile t0 + l (the launcher keeps runs below 64 tiles); clamped: the
    // tables of a malformed (flagged) batch may hold stale entries and must still stay in range
    const int ti = min(t0 + min(lane, t1 - t0), num_tiles);
    const int tf = min(max(tile_first[ti], 0), N), te = min(max(tile_edge[ti], 0), E);

    auto plan = [&](int ts) {
        FmStage st;
        st.ok = ts < t1 ? 1 : 0;
        st.nb = st.rows = st.e0 = st.ne = 0;
        st.next_t = ts;
        if (!st.ok)
            return st;
        const int rel = ts - t0;
        const int nb = __builtin_amdgcn_readlane(tf, rel), e0 = __builtin_amdgcn_readlane(te, rel);
        const unsigned long long fit = __ballot(lane > rel && lane <= t1 - t0 && tf - nb <= FM_CAP);
        st.nb = nb;
        st.e0 = e0;
        int endl = rel + 1; // (nothing fits: the next tile alone, cut to the stage -- only if the max_graph_nodes promise is broken)
        if (fit) {
            const unsigned long long nofit = ~fit & (~0ull << (rel + 1));
            endl = nofit ? __builtin_ctzll(nofit) - 1 : 63 - __builtin_clzll(fit);
        }
        st.rows = min(max(__builtin_amdgcn_readlane(tf, endl) - nb, 0), FM_CAP);
        st.ne = max(__builtin_amdgcn_readlane(te, endl) - e0, 0);
        st.next_t = t0 + endl;
        return st;
    };
    int vm = 0; // vector-memory instructions this wave has issued (DMA + stores): counted waits (VM operations retire in order)
    auto issue = [&](const FmStage &st, int bb) {
        if (!st.ok || st.rows <= 0)
            return;
        char *base = smem + (size_t)bb * in_b;
        const int nx = st.rows * F;
        for (int c = wave * 64; c < nx; c += FM_NW * 64, vm++)
            if (c + lane < nx)
                dma4_to_lds_u(x + (size_t)st.nb * F + c + lane, base + (size_t)c * 4);
        const char *grec = reinterpret_cast<const char *>(node_rec + 2 * (size_t)st.nb);
        const int rbytes = st.rows * 32;
        for (int c = ((wave + 2) & (FM_NW - 1)) * 1024; c < rbytes; c += FM_NW * 1024, vm++)
            if (c + lane * 16 < rbytes)
                dma16_to_lds_u(grec + c + lane * 16, base + rec_o + c);
        if (st.ne <= FM_ECAP)
            for (int c = ((wave + 4) & (FM_NW - 1)) * 64; c < st.ne; c += FM_NW * 64, vm++)
                if (c + lane < st.ne)
                    dma4_to_lds_u(col + st.e0 + c + lane, base + col_o + (size_t)c * 4);
    };

    FmStage cur = plan(t0);
    issue(cur, 0);
    int mark_cur = vm;

    // ---- wave roles: NS slices of 16 output columns; a wave owns two ADJACENT slices (32 columns) for the units rg, rg + RGN, ...
    const int NS = (Nout + 15) >> 4;
    const int SPW = NS >= 2 ? 2 : 1;
    int cwl = 0;
    while ((1 << cwl) * SPW < NS && cwl < 3)
        cwl++;
    const int CW = 1 << cwl, RGN = FM_NW >> cwl;
    const int cw = wave & (CW - 1), rg = wave >> cwl;
    // weight slices -> registers: k step t of block q multiplies stage column 16 q + lg + 4 t (A0 is stored to match)
    float wr[2][KQ * 4];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n = (cw * SPW + j) * 16 + li;
#pragma unroll
        for (int q = 0; q < KQ; q++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = 16 * q + lg + 4 * t;
                wr[j][q * 4 + t] = (j < SPW && n < Nout && k < K) ? W[(size_t)n * ldw + k] : 0.0f;
            }
    }
    float4 bq[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int c0 = (cw * SPW + j) * 16 + 4 * lg;
        bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && j < SPW) {
            bq[j].x = c0 + 0 < Nout ? bias[c0 + 0] : 0.f;
            bq[j].y = c0 + 1 < Nout ? bias[c0 + 1] : 0.f;
            bq[j].z = c0 + 2 < Nout ? bias[c0 + 2] : 0.f;
            bq[j].w = c0 + 3 < Nout ? bias[c0 + 3] : 0.f;
        }
    }
    // (tracked loads: finished HERE, or their first use inside the stage loop is guarded by a full vmcnt(0) -- k_stack.hip)
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int q = 0; q < KQ * 4; q++)
            asm volatile("" : "+v"(wr[j][q]));
        asm volatile("" : "+v"(bq[j].x), "+v"(bq[j].y), "+v"(bq[j].z), "+v"(bq[j].w));
    }
    const int nsteps = (K + 3) >> 2; // k steps that hold a feature
    const int G = 1 << glog2, RPI = 64 >> glog2; // lanes per output row (Nout / 4), rows per wave instruction

    int b = 0;
    while (cur.ok) {
        const int rows = cur.rows, nb = cur.nb;
        const char *base = smem + (size_t)b * in_b;
        const int4 *srec = reinterpret_cast<const int4 *>(base + rec_o);
        const int32_t *scol = reinterpret_cast<const int32_t *>(base + col_o);
        const bool col_lds = cur.ne <= FM_ECAP;
        // ---- the stage's inputs have landed (own share; then everybody's); everybody is done with A0, YT and the other buffer
        vmcnt_wait_n(min(vm - mark_cur, 63));
        g2_barrier();
        const FmStage nxt = plan(cur.next_t);
        issue(nxt, b ^ 1);
        const int mark_nxt = vm;

        // ---- P0: A0[i] = [mean_j x_j | x_i] in fragment order (k_conv_first's staged P0, MEAN + CAT)
        {
            constexpr int T0 = 2 * KQ; // columns per lane (16 KQ / 8)
            const float *xs = reinterpret_cast<const float *>(base);
            const int l8 = tid & 7;
            for (int i = tid >> 3; i < rows; i += FM_WG / 8) {
                const int4 r0 = srec[2 * i], r1 = srec[2 * i + 1];
                const int deg = r0.y;
                const int jg[4] = {r0.z, r0.w, r1.x, r1.y}; // batch-global ids; unused slots alias the row itself
                float acc[T0], xself[T0];
                int fcol[T0];
                bool own[T0];
#pragma unroll
                for (int t = 0; t < T0; t++) {
                    const int fk = l8 + 8 * t;
                    own[t] = fk >= F;
                    const int f = own[t] ? fk - F : fk;
                    fcol[t] = (fk < K && f < F) ? f : 0;
                    acc[t] = 0.0f;
                }
#pragma unroll
                for (int t = 0; t < T0; t++)
                    xself[t] = xs[i * F + fcol[t]];
                float xv[T0][4], c[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int jl = min(max(jg[q] - nb, 0), FM_CAP - 1);
                    c[q] = deg > q ? 1.0f : 0.0f;
#pragma unroll
                    for (int t = 0; t < T0; t++)
                        xv[t][q] = xs[jl * F + fcol[t]];
                }
#pragma unroll
                for (int t = 0; t < T0; t++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        acc[t] += xv[t][q] * c[q];
                if (deg > 4) { // the rest of the CSR row (two loops: a select between an LDS and a global pointer becomes a flat load)
                    auto more = [&](int j) {
                        const int jl = min(max(j - nb, 0), FM_CAP - 1);
#pragma unroll
                        for (int t = 0; t < T0; t++)
                            acc[t] += xs[jl * F + fcol[t]];
                    };
                    if (col_lds) {
                        for (int k = r0.x + 4; k < r0.x + deg; k++)
                            more(scol[min(max(k - cur.e0, 0), FM_ECAP - 1)]);
                    } else {
                        for (int k = r0.x + 4; k < r0.x + deg; k++)
                            more(col[k]);
                    }
                }
#pragma unroll
                for (int t = 0; t < T0; t++) {
                    const int fk = l8 + 8 * t;
                    const float v = own[t] ? xself[t] : (deg > 0 ? acc[t] * (1.0f / (float)deg) : 0.0f);
                    // stage column fk = 16 q + lg' + 4 t' sits at position 16 q + 4 lg' + t' (the fragment of k step t')
                    const int fp = (fk & ~15) | ((fk & 3) << 2) | ((fk >> 2) & 3);
                    A0[i * LD0 + fp] = fk < K ? v : 0.0f;
                }
            }
        }
        g2_barrier(); // A0 complete

        // ---- M: YT = act(A0 . W^T + b)
        {
            const int units = (rows + 15) >> 4;
            for (int u = rg; u < units; u += RGN) {
                const float *ap = A0 + (u * 16 + li) * LD0 + 4 * lg;
                float4 a4[KQ];
#pragma unroll
                for (int q = 0; q < KQ; q++)
                    a4[q] = *reinterpret_cast<const float4 *>(ap + 16 * q);
                f32x4 acc[2];
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[j] = (f32x4){bq[j].x, bq[j].y, bq[j].z, bq[j].w};
#pragma unroll
                for (int q = 0; q < KQ; q++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (q * 4 + t >= nsteps) // (wave-uniform: this k step holds no feature)
                            break;
                        const float av = t == 0 ? a4[q].x : (t == 1 ? a4[q].y : (t == 2 ? a4[q].z : a4[q].w));
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[0][q * 4 + t], av, acc[0], 0, 0, 0);
                        if (SPW == 2)
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[1][q * 4 + t], av, acc[1], 0, 0, 0);
                    }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int c0 = (cw * SPW + j) * 16 + 4 * lg;
                    // (FM_CAP = 56 is no multiple of 16: the last unit's rows 56 .. 63 lie past YT's carve -- not stored; A0's rows
                    // there are read from inside the allocation, YT's own first rows, and their products dropped here)
                    if (j < SPW && c0 < Nout && u * 16 + li < FM_CAP) // (Nout % 4 == 0: the lane's four columns are inside or outside together)
                        *reinterpret_cast<float4 *>(YT + (u * 16 + li) * LDY + c0) =
                            make_float4(act_t<ACT>(acc[j][0]), act_t<ACT>(acc[j][1]), act_t<ACT>(acc[j][2]), act_t<ACT>(acc[j][3]));
                }
            }
        }
        g2_barrier(); // the stage's output tile is complete

        // ---- OUT: every row to HBM, and the mean of its sources' rows (k_aggregate_ring<MEAN>: CSR order, one reciprocal per row)
        {
            typedef Vf<4> V;
            const int grp = lane >> glog2, gl = lane & (G - 1);
            const float *Yl = YT + gl * 4;
            for (int rb = wave * RPI; rb < rows; rb += FM_NW * RPI) {
                const int i = rb + grp;
                const bool active = i < rows;
                const int ic = active ? i : rb; // (lane groups past the stage re-read the pass's first row; their stores are predicated)
                const int4 r0 = srec[2 * ic], r1 = srec[2 * ic + 1];
                const int deg = r0.y;
                const int jl[4] = {r0.z - nb, r0.w - nb, r1.x - nb, r1.y - nb}; // unused slots alias the row itself
                const V y = V::load(Yl + ic * LDY);
                V h[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    h[q] = V::load(Yl + min(max(jl[q], 0), FM_CAP - 1) * LDY);
                // (the first term initialises the sum: 0 + v c, as the ring form starts from zero)
                V acc = vmul(h[0], V::splat(deg > 0 ? 1.0f : 0.0f));
                acc = vadd(acc, vmul(h[1], V::splat(deg > 1 ? 1.0f : 0.0f)));
                acc = vadd(acc, vmul(h[2], V::splat(deg > 2 ? 1.0f : 0.0f)));
                acc = vadd(acc, vmul(h[3], V::splat(deg > 3 ? 1.0f : 0.0f)));
                if (deg > 4) {
                    if (col_lds) {
                        for (int k = r0.x + 4; k < r0.x + deg; k++)
                            acc = vadd(acc, V::load(Yl + min(max(scol[min(max(k - cur.e0, 0), FM_ECAP - 1)] - nb, 0), FM_CAP - 1) * LDY));
                    } else {
                        for (int k = r0.x + 4; k < r0.x + deg; k++)
                            acc = vadd(acc, V::load(Yl + min(max(col[k] - nb, 0), FM_CAP - 1) * LDY));
                    }
                }
                const V m = deg > 0 ? vmul(acc, V::splat(1.0f / (float)deg)) : acc;
                vm += 2; // (the pass's first row exists: both store instructions have an active lane)
                if (active) {
                    agg_store<true>(y, Y + (size_t)(nb + i) * Nout + gl * 4);
                    agg_store<true>(m, Mo + (size_t)(nb + i) * Nout + gl * 4);
                }
            }
        }
        // (the 16-B stores are COUNTED in `vm` like the DMA: the wait at the top of the next stage must leave them in flight)
        cur = nxt;
        mark_cur = mark_nxt;
        b ^= 1;
    }
}

// hipErrorNotSupported (nothing launched): the caller runs k_conv_first and, for the next layer, the aggregate kernel
hipError_t launch_sage_first_mean(const BatchTables &t, const float *x, int F, const float *w, int ldw, const float *bias, float *y,
                                  float *mean_out, int Nout, int act, hipStream_t s)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    const int K = 2 * F;
    if (!options().sage_first_mean || F < 1 || K > 32 || !(Nout == 256 || Nout == 128 || Nout == 64) || t.tile_lo != 0 || (((uintptr_t)x) & 3))
        return hipErrorNotSupported;
    if (t.max_graph_nodes_hint <= 0 || t.max_graph_nodes_hint + t.tile_rows - 1 > FM_CAP)
        return hipErrorNotSupported; // whole graphs must fit a stage (validated on the device by graph prep: flag 8)
    // a batch with a large segment: the promise covers graphs [0, promise_graphs) only and graph prep validates nothing about the
    // rest -- those graphs need not fit a stage (round-5 advisor finding: they got clamped sources, unflagged): layer by layer
    if (t.promise_graphs < t.num_graphs || t.large_n >= 0)
        return hipErrorNotSupported;
    if ((((uintptr_t)y | (uintptr_t)mean_out) & 15))
        return hipErrorNotSupported;
    const int kq = K <= 16 ? 1 : 2;
    const int xs_b = ((FM_CAP * F * 4) + 15) & ~15;
    const size_t in_b = (size_t)xs_b + FM_CAP * 32 + FM_ECAP * 4;
    const size_t lds = 2 * in_b + (size_t)FM_CAP * (16 * kq + 4) * 4 + (size_t)FM_CAP * (Nout + 4) * 4;
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
    auto go = [&](auto atag, auto qtag) {
        constexpr int ACT = decltype(atag)::value, KQ = decltype(qtag)::value;
        auto kern = k_sage_first_mean<ACT, KQ>;
        if (ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds) != hipSuccess)
            return;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FM_WG), lds, s, x, F, t.node_rec, t.col, t.tile_first, t.tile_edge, t.num_tiles,
                           t.num_nodes, t.num_edges, w, ldw, bias, K, Nout, glog2, y, mean_out);
        rc = hipGetLastError();
    };
    auto go_a = [&](auto atag) {
        if (kq == 1)
            go(atag, IntTag<1>{});
        else
            go(atag, IntTag<2>{});
    };
    GNNB_DISPATCH_ACT(act, go_a)
    return rc;
}

} // namespace gnnb
