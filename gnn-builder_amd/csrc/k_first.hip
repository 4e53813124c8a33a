// k_first.hip -- a conv layer with a NARROW input (F_in <= 32: the first layer of every BASELINE model) in ring form:
// whole graphs staged in LDS, aggregate there, all output columns from one staged stage
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include "gnnb_stack.h"

namespace gnnb {

// =====================================================================================
// k_conv_first: out = act( aggregate(x) . W^T + b )  for a narrow x  (round 4)
// =====================================================================================
// Reference: the first conv layer of compute_gnn_head (templates/model.cpp.jinja:151-359) -- gcn_conv / gin_conv's first
// linear / sage_conv (gnn_builder_lib.h:1213-1387, :1389-1544, :2161-2341): per node "aggregate the neighbours' feature
// vectors, then `linear`".  The input rows are 36-128 bytes (QM9 11 floats, ogbg-molhiv 9), the output rows 512-1024
// bytes: the layer is bound by its OUTPUT stores (214 MB at BASELINE config 5) plus the narrow product.
//
// Round 3 ran it inside k_linear_reg's A stage (launch_conv_gather): every element of a 32-row stage gathered with five
// dependent global loads, once per 128-column block, the tracked gather loads of stage j + 1 retiring behind the output
// stores of stage j -- 91-95 us at config 5 against a ~36 us store floor.  Here, as in the conv-stack kernels:
//   DMA   x rows + node records + normalisers + CSR slice of a stage of WHOLE graphs -> LDS  (global_load_lds, two buffers)
//   P0    A0 = aggregate(x)  [rows, K <= 32]  eight lanes per row, LDS -> LDS            (GCN | SUM | MEAN, [mean | x] for SAGE)
//   M     out = act(A0 . W^T + b): v_mfma_f32_16x16x4_f32 with the operands swapped (a lane ends with four consecutive
//         columns of a row), every wave holds its 16- or 32-column slice of W in registers, ALL N <= 256 columns from the
//         one staged A0; k steps that hold no feature are skipped (k = lg + 4 t: 18 features = five steps, not eight)
//   ST    16-B stores straight from the accumulators (a wave's two slices are adjacent: 128 B per row)
// Graphs larger than a stage (no max_graph_nodes promise on this path) are taken in pieces whose sources are read from
// global memory (L2) instead of LDS -- same arithmetic, same order.
// Sums run in CSR order with the self term last, as k_aggregate_ring and the reference do.
#ifndef F1_CAP_ROWS
#define F1_CAP_ROWS 128
#endif
#ifndef F1_ABLATE   // development: 1 no stores, 2 no MFMA, 4 no P0, 8 no DMA (timing only: WRONG results)
#define F1_ABLATE 0
#endif
#ifndef F1_NT_STORE
#define F1_NT_STORE 1
#endif
static constexpr int F1_NW = 8, F1_WG = F1_NW * 64, F1_CAP = F1_CAP_ROWS, F1_ECAP = 8 * F1_CAP_ROWS; // rows / CSR entries per stage

struct F1Stage {
    int ok, nb, rows, e0, ne, direct, next_t, next_row; // direct: the piece's sources are read from global memory
};

template <int ACT, int KQ, int MODE, bool CAT>
__global__ __launch_bounds__(F1_WG, 2) void k_conv_first(
    const float *__restrict__ x, int F, const int4 *__restrict__ node_rec, const int32_t *__restrict__ col,
    const float *__restrict__ dinv, const int32_t *__restrict__ tile_first, const int32_t *__restrict__ tile_edge,
    int num_tiles, int N, int E, const float *__restrict__ W, int ldw, const float *__restrict__ bias, int K, int Nout,
    float eps, float *__restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD0 = 16 * KQ + 4; // A0 row (floats), padded: conflict-free fragment reads
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // ---- LDS carve: two input buffers {x rows | records | dinv | CSR slice}, one A0
    const int xs_b = ((F1_CAP * F * 4) + 15) & ~15;
    const int rec_o = xs_b, dinv_o = rec_o + F1_CAP * 32, col_o = dinv_o + F1_CAP * 4, in_b = col_o + F1_ECAP * 4;
    float *A0 = reinterpret_cast<float *>(smem + 2 * (size_t)in_b);
    // per-wave scratch for the output transpose: 16 rows x 32 columns (+ 4 floats of padding per row)
    float *ST = A0 + F1_CAP * LD0 + wave * (16 * 36);

    int t0, t1;

    run_cuts(blockIdx.x, gridDim.x, (unsigned)num_tiles, t0, t1); // (32-bit: gnnb_device.h)
    if (t1 <= t0)
        return;
    // window of the tile table in registers: lane l holds tile t0 + l (the launcher keeps runs below 64 tiles)
    const int ti = min(t0 + min(lane, t1 - t0), num_tiles);
    const int tf = min(max(tile_first[ti], 0), N), te = min(max(tile_edge[ti], 0), E);

    // ---- stage plan: the longest run of whole tiles from tile `ts` that fits the stage (rows and CSR slice); a tile that does
    // not fit alone -- one graph beyond the stage -- goes in direct pieces of F1_CAP rows
    auto plan = [&](int ts, int row_at) {
        F1Stage st;
        st.ok = ts < t1 ? 1 : 0;
        st.nb = st.rows = st.e0 = st.ne = st.direct = 0;
        st.next_t = ts;
        st.next_row = row_at;
        if (!st.ok)
            return st;
        const int rel = ts - t0;
        const int nb = __builtin_amdgcn_readlane(tf, rel), e0 = __builtin_amdgcn_readlane(te, rel);
        if (row_at > nb) { // inside an oversize tile: the next piece
            const int tile_end = __builtin_amdgcn_readlane(tf, rel + 1);
            st.nb = row_at;
            st.rows = min(tile_end - row_at, F1_CAP);
            st.direct = 1;
            st.next_row = row_at + st.rows;
            if (st.next_row >= tile_end) {
                st.next_t = ts + 1;
                st.next_row = tile_end;
            }
            return st;
        }
        const unsigned long long fit = __ballot(lane > rel && lane <= t1 - t0 && tf - nb <= F1_CAP && te - e0 <= F1_ECAP && te >= e0);
        st.nb = nb;
        st.e0 = e0;
        if (fit == 0) { // the tile alone exceeds the stage
            const int tile_end = __builtin_amdgcn_readlane(tf, rel + 1);
            st.rows = min(tile_end - nb, F1_CAP);
            st.direct = 1;
            st.next_row = nb + st.rows;
            if (st.next_row >= tile_end)
                st.next_t = ts + 1;
            return st;
        }
        // (the feasible ends are a prefix of the lanes behind `rel`: the last of them)
        const unsigned long long nofit = ~fit & (~0ull << (rel + 1));
        const int endl = nofit ? __builtin_ctzll(nofit) - 1 : 63 - __builtin_clzll(fit);
        st.rows = max(__builtin_amdgcn_readlane(tf, endl) - nb, 0);
        st.ne = max(__builtin_amdgcn_readlane(te, endl) - e0, 0);
        st.next_t = t0 + endl;
        st.next_row = nb + st.rows;
        return st;
    };
    int vm = 0; // vector-memory instructions this wave has issued (DMA + stores): counted waits (VM operations retire in order)
    auto issue = [&](const F1Stage &st, int bb) {
        if (!st.ok || st.direct || st.rows <= 0 || (F1_ABLATE & 8))
            return;
        char *base = smem + (size_t)bb * in_b;
        const int nx = st.rows * F;
        for (int c = wave * 64; c < nx; c += F1_NW * 64, vm++)
            if (c + lane < nx)
                dma4_to_lds_u(x + (size_t)st.nb * F + c + lane, base + (size_t)c * 4);
        const char *grec = reinterpret_cast<const char *>(node_rec + 2 * (size_t)st.nb);
        const int rbytes = st.rows * 32;
        for (int c = ((wave + 2) & (F1_NW - 1)) * 1024; c < rbytes; c += F1_NW * 1024, vm++)
            if (c + lane * 16 < rbytes)
                dma16_to_lds_u(grec + c + lane * 16, base + rec_o + c);
        if (MODE == GNNB_AGG_GCN)
            for (int c = ((wave + 4) & (F1_NW - 1)) * 64; c < st.rows; c += F1_NW * 64, vm++)
                if (c + lane < st.rows)
                    dma4_to_lds_u(dinv + st.nb + c + lane, base + dinv_o + (size_t)c * 4);
        for (int c = ((wave + 6) & (F1_NW - 1)) * 64; c < st.ne; c += F1_NW * 64, vm++)
            if (c + lane < st.ne)
                dma4_to_lds_u(col + st.e0 + c + lane, base + col_o + (size_t)c * 4);
    };

    F1Stage cur = plan(t0, 0);
    issue(cur, 0);
    int mark_cur = vm;
    F1Stage nxt = plan(cur.next_t, cur.next_row);
    issue(nxt, 1);
    int mark_nxt = vm;

    // ---- wave roles: NS = slices of 16 output columns; a wave owns SPW adjacent slices (32 columns) for the units rg,
    // rg + RGN, ...
    const int NS = (Nout + 15) >> 4;
    const int SPW = NS >= 2 ? 2 : 1; // (two ADJACENT slices per wave wherever there are two: whole 128-B lines per stored row)
    int cwl = 0;
    while ((1 << cwl) * SPW < NS && cwl < 3)
        cwl++;
    const int CW = 1 << cwl, RGN = F1_NW >> cwl;
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
    const bool vec_out = (Nout % 4 == 0) && (((uintptr_t)Y & 15) == 0);

    int b = 0;
    while (cur.ok) {
        const int rows = cur.rows, nb = cur.nb;
        // ---- the stage's inputs have landed (own share; then everybody's) and everybody is done with A0
        if (!cur.direct)
            vmcnt_wait_n(min(vm - mark_cur, 63));
        g2_barrier();
        // ---- P0: A0[i][pos(fk)] for the stage column fk < K: eight lanes per row, lane l8 takes columns l8, l8 + 8, ...
        // TWO copies of the loop, staged and direct, chosen per stage: inside one loop the compiler joins the two sources of
        // a value behind an unconditional s_waitcnt vmcnt(0) -- the direct path's global loads -- which in EVERY row pass
        // also waited for the DMA of the stages ahead and the output stores of the stage before (77 us instead of 4x)
        auto p0 = [&](auto dtag) {
            constexpr bool DIRECT = decltype(dtag)::value != 0;
            constexpr int T0 = 2 * KQ; // columns per lane (16 KQ / 8)
            const char *base = smem + (size_t)b * in_b;
            const float *xs = reinterpret_cast<const float *>(base);
            const int4 *srec = reinterpret_cast<const int4 *>(base + rec_o);
            const float *sdinv = reinterpret_cast<const float *>(base + dinv_o);
            const int32_t *scol = reinterpret_cast<const int32_t *>(base + col_o);
            const int l8 = tid & 7;
            for (int i = tid >> 3; i < rows; i += F1_WG / 8) {
                int4 r0, r1;
                if (DIRECT) {
                    r0 = node_rec[2 * (size_t)(nb + i)];
                    r1 = node_rec[2 * (size_t)(nb + i) + 1];
                } else {
                    r0 = srec[2 * i];
                    r1 = srec[2 * i + 1];
                }
                const int deg = r0.y;
                const int jg[4] = {r0.z, r0.w, r1.x, r1.y}; // batch-global ids; unused slots alias the row itself
                float acc[T0], xself[T0];
                int fcol[T0];
                bool own[T0];
#pragma unroll
                for (int t = 0; t < T0; t++) {
                    const int fk = l8 + 8 * t;
                    own[t] = CAT && fk >= F;
                    const int f = own[t] ? fk - F : fk;
                    fcol[t] = (fk < K && f < F) ? f : 0;
                    acc[t] = 0.0f;
                }
                float di = 1.0f;
                if (DIRECT) {
                    if (MODE == GNNB_AGG_GCN)
                        di = dinv[nb + i];
#pragma unroll
                    for (int t = 0; t < T0; t++)
                        xself[t] = x[(size_t)(nb + i) * F + fcol[t]];
                    for (int k = 0; k < deg; k++) {
                        const int j = k < 4 ? jg[k] : col[r0.x + k];
                        const float cj = MODE == GNNB_AGG_GCN ? di * dinv[j] : 1.0f;
#pragma unroll
                        for (int t = 0; t < T0; t++)
                            acc[t] += x[(size_t)j * F + fcol[t]] * cj;
                    }
                } else {
                    if (MODE == GNNB_AGG_GCN)
                        di = sdinv[i];
#pragma unroll
                    for (int t = 0; t < T0; t++)
                        xself[t] = xs[i * F + fcol[t]];
                    float xv[T0][4], c[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int jl = jg[q] - nb;
                        c[q] = deg > q ? (MODE == GNNB_AGG_GCN ? di * sdinv[jl] : 1.0f) : 0.0f;
#pragma unroll
                        for (int t = 0; t < T0; t++)
                            xv[t][q] = xs[jl * F + fcol[t]];
                    }
#pragma unroll
                    for (int t = 0; t < T0; t++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            acc[t] += xv[t][q] * c[q];
                    for (int k = r0.x + 4; k < r0.x + deg; k++) { // degree > 4: the rest of the CSR row, from the staged slice
                        const int jl = scol[min(max(k - cur.e0, 0), F1_ECAP - 1)] - nb;
                        const float cj = MODE == GNNB_AGG_GCN ? di * sdinv[jl] : 1.0f;
#pragma unroll
                        for (int t = 0; t < T0; t++)
                            acc[t] += xs[jl * F + fcol[t]] * cj;
                    }
                }
#pragma unroll
                for (int t = 0; t < T0; t++) {
                    const int fk = l8 + 8 * t;
                    float v;
                    if (own[t])
                        v = xself[t];
                    else if (MODE == GNNB_AGG_GCN)
                        v = acc[t] + xself[t] * (di * di);
                    else if (MODE == GNNB_AGG_SUM)
                        v = acc[t] + xself[t] * (1.0f + eps);
                    else
                        v = deg > 0 ? acc[t] * (1.0f / (float)deg) : 0.0f; // (one reciprocal per row, as k_aggregate_ring)
                    // stage column fk = 16 q + lg' + 4 t' sits at position 16 q + 4 lg' + t' (the fragment of k step t')
                    const int fp = (fk & ~15) | ((fk & 3) << 2) | ((fk >> 2) & 3);
                    A0[i * LD0 + fp] = fk < K ? v : 0.0f;
                }
            }
        };
        if (F1_ABLATE & 4) {
        } else if (cur.direct)
            p0(IntTag<1>{});
        else
            p0(IntTag<0>{});
        g2_barrier(); // A0 complete; the input buffer is free
        // ---- the stage after next starts its way to LDS (into the buffer P0 just consumed)
        const F1Stage nn = plan(nxt.next_t, nxt.next_row);
        issue(nn, b);
        const int mark_nn = vm;

        // ---- M + ST
        {
            const int units = (rows + 15) >> 4;
            // 16-B store instructions this wave issues per unit: one per slice it owns that holds a column (wave-uniform;
            // such an instruction always has an active lane -- row u * 16 of the unit, columns 4 lg = 0 of the slice)
            const int st_per_unit = vec_out ? min(max(NS - cw * SPW, 0), SPW) : 0;
            const bool wide_rows = vec_out && SPW == 2 && cw * 32 + 32 <= Nout; // (wave-uniform)
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
                        if (q * 4 + t >= nsteps || (F1_ABLATE & 2)) // (wave-uniform: this k step holds no feature)
                            break;
                        const float av = t == 0 ? a4[q].x : (t == 1 ? a4[q].y : (t == 2 ? a4[q].z : a4[q].w));
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[0][q * 4 + t], av, acc[0], 0, 0, 0);
                        if (SPW == 2)
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[1][q * 4 + t], av, acc[1], 0, 0, 0);
                    }
                const int row = u * 16 + li;
                if (wide_rows) {
                    // two adjacent slices = 32 columns = one 128-B line per row: through the wave's LDS scratch, so that a
                    // store instruction writes EIGHT WHOLE lines (8 lanes x 16 B per row) instead of sixteen 64-B halves --
                    // straight from the accumulators the kernel was store-bound at 2.9 TB/s (77 us at BASELINE config 5,
                    // 18 us with the stores compiled out)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        *reinterpret_cast<float4 *>(ST + li * 36 + j * 16 + 4 * lg) =
                            make_float4(act_t<ACT>(acc[j][0]), act_t<ACT>(acc[j][1]), act_t<ACT>(acc[j][2]), act_t<ACT>(acc[j][3]));
                    // (same wave writes and reads: the LDS executes a wave's operations in order)
                    vm += u * 16 + 8 < rows ? 2 : 1; // (a store instruction per half of the unit that holds a row)
#pragma unroll
                    for (int ps = 0; ps < 2; ps++) {
                        const int rr = ps * 8 + (lane >> 3), cc = (lane & 7) * 4;
                        const float4 v = *reinterpret_cast<const float4 *>(ST + rr * 36 + cc);
                        if (u * 16 + rr < rows && !(F1_ABLATE & 1)) {
                            float *yp = Y + (size_t)(nb + u * 16 + rr) * Nout + cw * 32 + cc;
#if F1_NT_STORE
                            agg_f32x4 tv = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(tv, reinterpret_cast<agg_f32x4 *>(yp));
#else
                            *reinterpret_cast<float4 *>(yp) = v;
#endif
                        }
                    }
                    continue;
                }
                vm += st_per_unit;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (j >= SPW)
                        break;
                    const int c0 = (cw * SPW + j) * 16 + 4 * lg;
                    float4 v = make_float4(act_t<ACT>(acc[j][0]), act_t<ACT>(acc[j][1]), act_t<ACT>(acc[j][2]), act_t<ACT>(acc[j][3]));
                    float *yp = Y + (size_t)(nb + row) * Nout + c0;
                    if (row < rows && c0 < Nout && !(F1_ABLATE & 1)) {
                        if (vec_out && c0 + 3 < Nout) {
#if F1_NT_STORE
                            agg_f32x4 tv = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(tv, reinterpret_cast<agg_f32x4 *>(yp));
#else
                            *reinterpret_cast<float4 *>(yp) = v;
#endif
                        } else {
                            yp[0] = v.x;
                            if (c0 + 1 < Nout) yp[1] = v.y;
                            if (c0 + 2 < Nout) yp[2] = v.z;
                            if (c0 + 3 < Nout) yp[3] = v.w;
                        }
                    }
                }
            }
        }
        // (the 16-B stores are COUNTED in `vm` like the DMA: the wait at the top of the next stage must leave them -- and the
        // DMA of the stage after next, issued in front of them -- in flight; uncounted, every stage waited for a memory round
        // trip: 77 us instead of 4x us at BASELINE config 5.  The scalar-store path is not counted: its waits are stricter.)
        cur = nxt;
        mark_cur = mark_nxt;
        nxt = nn;
        mark_nxt = mark_nn;
        b ^= 1;
    }
}

// ---------------------------------------------------------------------------------------
// hipErrorNotSupported (nothing launched): the caller takes the k_linear_reg gather form
hipError_t launch_conv_first(const BatchTables &t, int agg_kind, float eps, const float *x, int F, int K, const float *w,
                             int ldw, const float *bias, float *y, int Nout, int act, hipStream_t s, int cat)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    if (!(agg_kind == GNNB_AGG_GCN || agg_kind == GNNB_AGG_SUM || agg_kind == GNNB_AGG_MEAN) || F < 1 || K > 32 || K < 1 ||
        Nout < 1 || Nout > 256 || (cat > 0 && (cat != F || K != 2 * F || agg_kind != GNNB_AGG_MEAN)) || (cat == 0 && K != F) ||
        t.tile_lo != 0 || (((uintptr_t)x) & 3))
        return hipErrorNotSupported;
    const int kq = K <= 16 ? 1 : 2;
    const int xs_b = ((F1_CAP * F * 4) + 15) & ~15;
    const size_t in_b = (size_t)xs_b + F1_CAP * 32 + F1_CAP * 4 + F1_ECAP * 4;
    const size_t lds = 2 * in_b + (size_t)F1_CAP * (16 * kq + 4) * 4 + (size_t)F1_NW * 16 * 36 * 4;
    const int cus = device_cu_count();
    long long grid = std::min<long long>(2LL * cus, t.num_tiles);
    if (grid < 1)
        grid = 1;
    if ((t.num_tiles + grid - 1) / grid > 62) // a workgroup keeps its run of the tile table in one register per lane
        grid = (t.num_tiles + 61) / 62;
    hipError_t rc = hipErrorNotSupported;
    auto go = [&](auto atag, auto qtag, auto mtag, auto ctag) {
        constexpr int ACT = decltype(atag)::value, KQ = decltype(qtag)::value, MODE = decltype(mtag)::value;
        constexpr bool CAT = decltype(ctag)::value != 0;
        auto kern = k_conv_first<ACT, KQ, MODE, CAT>;
        if (ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds) != hipSuccess)
            return;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(F1_WG), lds, s, x, F, t.node_rec, t.col, t.dinv, t.tile_first, t.tile_edge,
                           t.num_tiles, t.num_nodes, t.num_edges, w, ldw, bias, K, Nout, eps, y);
        rc = hipGetLastError();
    };
    auto go_a = [&](auto atag) {
        if (agg_kind == GNNB_AGG_GCN) {
            if (kq == 1) go(atag, IntTag<1>{}, IntTag<GNNB_AGG_GCN>{}, IntTag<0>{});
            else go(atag, IntTag<2>{}, IntTag<GNNB_AGG_GCN>{}, IntTag<0>{});
        } else if (agg_kind == GNNB_AGG_SUM) {
            if (kq == 1) go(atag, IntTag<1>{}, IntTag<GNNB_AGG_SUM>{}, IntTag<0>{});
            else go(atag, IntTag<2>{}, IntTag<GNNB_AGG_SUM>{}, IntTag<0>{});
        } else if (cat > 0) {
            if (kq == 1) go(atag, IntTag<1>{}, IntTag<GNNB_AGG_MEAN>{}, IntTag<1>{});
            else go(atag, IntTag<2>{}, IntTag<GNNB_AGG_MEAN>{}, IntTag<1>{});
        } else {
            if (kq == 1) go(atag, IntTag<1>{}, IntTag<GNNB_AGG_MEAN>{}, IntTag<0>{});
            else go(atag, IntTag<2>{}, IntTag<GNNB_AGG_MEAN>{}, IntTag<0>{});
        }
    };
    GNNB_DISPATCH_ACT(act, go_a)
    return rc;
}

} // namespace gnnb
