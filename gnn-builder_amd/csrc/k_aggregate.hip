// k_aggregate.hip -- gather -> segmented reduce per destination row (HBM bound: the roofline kernel)
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include "gnnb_device.h"

namespace gnnb {

// =====================================================================================
// gather-aggregate
// =====================================================================================
// One workgroup owns a run of node tiles, i.e. a few WHOLE graphs (tiles are cut at graph
// boundaries), so every neighbour row a destination needs lies inside the workgroup's own
// node range.  The node rows are streamed from HBM exactly once, fully coalesced (16 B per
// lane), into LDS; the CSR slice (row_ptr, col) of the tile is staged next to them.  Each
// destination row is then reduced by a lane group (width/4 lanes, float4 per lane) reading
// its neighbours from LDS in CSR order -- no atomics, one owner per output row -- and the
// result is written back with 16-B coalesced stores.  Algorithmic HBM traffic per launch:
// 4*w*N read + 4*w*N*k_out write + 4*(N+1) + 4*E + 4*(T+1).
//
// A tile that does not fit the LDS budget (a graph far larger than tile_rows) takes the
// same code path with the neighbour rows read straight from global memory (L2).
//
// Semantics per mode:
//   GCN  gcn_conv_agg   gnn_builder_lib.h:1213-1289   sum_j x_j/sqrt((1+d_i)(1+d_j)) + x_i/sqrt((1+d_i)^2)
//   SUM  gin_conv_agg   gnn_builder_lib.h:1389-1437 + :1525-1535   sum_j x_j + x_i (1+eps)
//   MEAN sage_conv_agg  gnn_builder_lib.h:2161-2209   (sum_j x_j)/d, 0 when d = 0
//   PNA  pna_conv_agg   gnn_builder_lib.h:1750-1834 with h_ij = q_i + p_j (the per-edge
//        W_pre [x_i || x_j] + b split into two per-node products) and PyG's std:
//        sqrt(max(E[h^2]-E[h]^2, 1e-5)) zeroed where <= sqrt(1e-5)  (SURVEY finding 5).
// Neighbours are summed in CSR (= stable COO) order, the self term last, as the reference does.


// number of rows an output row has (PNA: max | min | mean | std)
template <int MODE>
struct AggOut {
    static constexpr int K = MODE == GNNB_AGG_PNA ? 4 : 1;
};

// The reduction itself, shared by the LDS-staged and the direct forms: `take` one neighbour row at a
// time in CSR order, `done` adds the self term / finalises and stores.
template <int MODE, int VEC, bool NT>
struct AggAcc {
    typedef Vf<VEC> V;
    V acc, vmx, vmn, s2;
    __device__ inline void init() { acc = V::splat(0.0f); vmx = acc; vmn = acc; s2 = acc; }
    // coef: GCN dinv_i * dinv_j, LG 1/sqrt(d_i d_j); xi: PNA's per-destination term q_i
    __device__ inline void take(const V &v, float coef, const V &xi, bool first)
    {
        if (MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_LG) {
            acc = vadd(acc, vmul(v, V::splat(coef)));
        } else if (MODE == GNNB_AGG_PNA) {
            const V h = vadd(xi, v);
            if (first) {
                vmx = h;
                vmn = h;
            } else {
                vmx = vmax(vmx, h);
                vmn = vmin(vmn, h);
            }
            acc = vadd(acc, h);
            s2 = vadd(s2, vmul(h, h));
        } else {
            acc = vadd(acc, v);
        }
    }
    __device__ inline void done(const V &xi, float di, int deg, float eps, float *__restrict__ out, size_t node, int w, int fo)
    {
        if (MODE == GNNB_AGG_GCN) {
            agg_store<NT>(vadd(acc, vmul(xi, V::splat(di * di))), out + node * w + fo);
        } else if (MODE == GNNB_AGG_SUM) {
            agg_store<NT>(vadd(acc, vmul(xi, V::splat(1.0f + eps))), out + node * w + fo);
        } else if (MODE == GNNB_AGG_MEAN) {
            // one reciprocal per row instead of a division per component (<= 1 ulp from sum / count)
            agg_store<NT>(deg > 0 ? vmul(acc, V::splat(1.0f / (float)deg)) : acc, out + node * w + fo);
        } else if (MODE == GNNB_AGG_PNA) {
            V mean = V::splat(0.0f), sd = V::splat(0.0f);
            if (deg > 0) {
                const V dn = V::splat((float)deg);
                mean = vdiv(acc, dn);
                sd = pyg_std(vdiv(s2, dn), mean);
            }
            float *o = out + node * 4 * w + fo;
            agg_store<NT>(vmx, o);
            agg_store<NT>(vmn, o + w);
            agg_store<NT>(mean, o + 2 * (size_t)w);
            agg_store<NT>(sd, o + 3 * (size_t)w);
        } else if (MODE == GNNB_AGG_COPY) {
            agg_store<NT>(xi, out + node * w + fo);
        } else { // LG, SIMPLE: no self term
            agg_store<NT>(acc, out + node * w + fo);
        }
    }
};

// One destination row reduced from an LDS stage that holds its whole graph: rows `sx`, node records
// `srec`, GCN normalisers `sdinv`, and (PNA, QLDS) the per-destination terms `sq`, all indexed by
// row - nb.  begin() issues every LDS read of the row (its record gives the first four sources, unused
// slots alias the row itself), finish() reduces and stores: two rows are begun before either is
// finished so that ten ds_read_b128 are in flight per lane.
template <int MODE, int VEC, bool QLDS = false, bool NT = false>
struct LdsRow {
    typedef Vf<VEC> V;
    int node, rp0, deg, jr[4];
    float di, sj[4];
    V xi, nbv[4];
    bool valid;

    __device__ inline void begin(bool ok, int nb, int r_, const float *sx, const float *sq, const int4 *srec,
                                 const float *sdinv, const float *__restrict__ selfq, int w, int fo)
    {
        valid = ok;
        if (!ok)
            return;
        node = nb + r_;
        if (MODE == GNNB_AGG_COPY) {
            xi = V::load(sx + (size_t)r_ * w + fo);
            return;
        }
        const int4 r0 = srec[2 * r_], r1 = srec[2 * r_ + 1];
        rp0 = r0.x;
        deg = r0.y;
        jr[0] = r0.z - nb;
        jr[1] = r0.w - nb;
        jr[2] = r1.x - nb;
        jr[3] = r1.y - nb;
        if (MODE == GNNB_AGG_PNA)
            xi = selfq == nullptr ? V::splat(0.0f) // (no destination term: PNA's degree-class form folds it into the post-NN weights)
                                  : (QLDS ? V::load(sq + (size_t)r_ * w + fo) : V::load(selfq + (size_t)node * w + fo));
        else if (MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_SUM)
            xi = V::load(sx + (size_t)r_ * w + fo);
#pragma unroll
        for (int q = 0; q < 4; q++)
            nbv[q] = V::load(sx + (size_t)jr[q] * w + fo); // unused slots alias the row itself
        if (MODE == GNNB_AGG_GCN) {
            di = sdinv[r_];
#pragma unroll
            for (int q = 0; q < 4; q++)
                sj[q] = di * sdinv[jr[q]];
        } else if (MODE == GNNB_AGG_LG) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                sj[q] = lg_coef(deg, srec[2 * jr[q]].y);
        }
    }

    __device__ inline void finish(int nb, const float *sx, const int4 *srec, const float *sdinv,
                                  const int32_t *__restrict__ col, float *__restrict__ out, int w, int fo,
                                  float eps)
    {
        if (!valid)
            return;
        AggAcc<MODE, VEC, NT> a;
        a.init();
        if (MODE != GNNB_AGG_COPY) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (deg > q)
                    a.take(nbv[q], sj[q], xi, q == 0);
            for (int k = rp0 + 4; k < rp0 + deg; k++) { // degree > 4: the rest of the CSR row
                const int j = col[k] - nb;
                const V v = V::load(sx + (size_t)j * w + fo);
                float c = 0.0f;
                if (MODE == GNNB_AGG_GCN)
                    c = di * sdinv[j];
                else if (MODE == GNNB_AGG_LG)
                    c = lg_coef(deg, srec[2 * j].y);
                a.take(v, c, xi, false);
            }
        }
        a.done(xi, di, deg, eps, out, (size_t)node, w, fo);
    }
};

// Lean form of LdsRow for the float4 modes GCN, SUM, MEAN and SIMPLE, written for instruction count: the reduce phase of
// the ring kernel is bound by vector / scalar issue (round 2: 46 vector and 58 scalar instructions per output row).
// Differences to LdsRow: unused neighbour slots get coefficient 0 instead of a select per value (4 selects on scalars
// instead of 16 on vector components); LDS addresses are one v_mad_u32_u24 per neighbour on bases that already hold the
// stage's -nb offset (no per-neighbour subtraction, no 32-bit multiply); the output address is a 32-bit lane offset on
// the stage's base pointer (no 64-bit vector arithmetic per store).  Same sums in the same order as LdsRow / AggAcc.
typedef __attribute__((address_space(3))) const agg_f32x4 *agg_lds_f4;
typedef __attribute__((address_space(3))) const float *agg_lds_f;
__device__ __forceinline__ float4 agg_lds_ld4(uint32_t addr) // ds_read_b128 at an LDS byte address
{
    const agg_f32x4 t = *(agg_lds_f4)(uintptr_t)addr;
    return make_float4(t.x, t.y, t.z, t.w);
}
typedef __attribute__((address_space(3))) const int32_t *agg_lds_i;
template <int MODE, bool NT>
struct LdsRowF {
    typedef Vf<4> V;
    int rp0, deg;
    uint32_t roff;
    float di, c[4];
    V xi, nbv[4];
    bool valid;

    // sxo  = LDS byte address of the stage's rows minus nb * w4, plus this lane's column offset
    // sdo  = LDS byte address of its normalisers minus nb * 4 (GCN)
    __device__ __forceinline__ void begin(bool ok, int r_, const int4 *srec, uint32_t sxo, uint32_t sx_lane,
                                          const float *sdinv, const float4 *sgc, int w4)
    {
        // (lane groups without a row read the stage's FIRST row and only their store is predicated: a predicated begin
        // merges every loaded register with its previous value behind the branch -- ~16 v_mov per row in the chain
        // between the LDS reads and the FMAs)
        valid = ok;
        r_ = ok ? r_ : 0;
        const int4 r0 = srec[2 * r_], r1 = srec[2 * r_ + 1];
        rp0 = r0.x;
        deg = r0.y;
        roff = (uint32_t)__mul24(r_, w4);
        const int jg[4] = {r0.z, r0.w, r1.x, r1.y}; // batch-global ids; unused slots alias the row itself
        if (MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_SUM)
            xi.v = agg_lds_ld4(sx_lane + roff);
#pragma unroll
        for (int q = 0; q < 4; q++)
            nbv[q].v = agg_lds_ld4(sxo + (uint32_t)__mul24(jg[q], w4));
        if (MODE == GNNB_AGG_GCN) {
            di = sdinv[r_];
            const float4 g = sgc[r_]; // dinv_i dinv_j of the inline neighbours, 0 past the degree (k_gcn_coef)
            c[0] = g.x, c[1] = g.y, c[2] = g.z, c[3] = g.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                c[q] = deg > q ? 1.0f : 0.0f;
        }
    }

    // scol_o = LDS byte address of the stage's CSR slice minus e0 * 4; obase = the stage's first output row (+ column)
    __device__ __forceinline__ void finish(uint32_t sxo, uint32_t sdo, uint32_t scol_o, int w4, char *__restrict__ obase, float eps)
    {
        V acc;
        // (the first term initialises the sum: 0 + v c, as AggAcc starts from zero)
        acc = vmul(nbv[0], V::splat(c[0]));
        acc = vadd(acc, vmul(nbv[1], V::splat(c[1])));
        acc = vadd(acc, vmul(nbv[2], V::splat(c[2])));
        acc = vadd(acc, vmul(nbv[3], V::splat(c[3])));
        for (int k = rp0 + 4; k < rp0 + deg; k++) { // degree > 4: the rest of the CSR row, from the stage's slice in LDS
            const int j = *(agg_lds_i)(uintptr_t)(scol_o + (uint32_t)k * 4u);
            V v;
            v.v = agg_lds_ld4(sxo + (uint32_t)__mul24(j, w4));
            const float cj = MODE == GNNB_AGG_GCN ? di * *(agg_lds_f)(uintptr_t)(sdo + (uint32_t)j * 4u) : 1.0f;
            acc = vadd(acc, vmul(v, V::splat(cj)));
        }
        V o;
        if (MODE == GNNB_AGG_GCN)
            o = vadd(acc, vmul(xi, V::splat(di * di)));
        else if (MODE == GNNB_AGG_SUM)
            o = vadd(acc, vmul(xi, V::splat(1.0f + eps)));
        else if (MODE == GNNB_AGG_MEAN)
            o = deg > 0 ? vmul(acc, V::splat(1.0f / (float)deg)) : acc;
        else
            o = acc;
        if (valid)
            agg_store<NT>(o, reinterpret_cast<float *>(obase + roff));
    }
};

// The same row straight from global memory (a graph larger than an LDS stage): neighbour rows are
// L2-side gathers, the degree comes from the node record (row_ptr holds row STARTS only: dropped edges
// leave gaps at the end of a graph's CSR segment, so row_ptr[v+1] - row_ptr[v] is not a degree).
template <int MODE, int VEC, bool NT>
__device__ inline void agg_row_direct(int node, const float *__restrict__ x, const float *__restrict__ selfq,
                                      float *__restrict__ out, const int4 *__restrict__ node_rec,
                                      const int32_t *__restrict__ col, const float *__restrict__ dinv, int w, int fo,
                                      float eps)
{
    typedef Vf<VEC> V;
    AggAcc<MODE, VEC, NT> a;
    a.init();
    int rp0 = 0, deg = 0;
    if (MODE != GNNB_AGG_COPY) {
        const int4 r0 = node_rec[2 * (size_t)node];
        rp0 = r0.x;
        deg = r0.y;
    }
    const float di = (MODE == GNNB_AGG_GCN) ? dinv[node] : 0.0f;
    V xi = V::splat(0.0f);
    if (MODE == GNNB_AGG_PNA)
        xi = selfq == nullptr ? V::splat(0.0f) : V::load(selfq + (size_t)node * w + fo);
    else if (MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_SUM || MODE == GNNB_AGG_COPY)
        xi = V::load(x + (size_t)node * w + fo);
    for (int k = rp0; k < rp0 + deg; k++) {
        const int j = col[k];
        const V xj = V::load(x + (size_t)j * w + fo);
        float c = 0.0f;
        if (MODE == GNNB_AGG_GCN)
            c = di * dinv[j];
        else if (MODE == GNNB_AGG_LG)
            c = lg_coef(deg, node_rec[2 * (size_t)j].y);
        a.take(xj, c, xi, k == rp0);
    }
    a.done(xi, di, deg, eps, out, (size_t)node, w, fo);
}

// -------------------------------------------------------------------------------------
// Ring form: persistent, ONE workgroup of up to 16 waves per CU, whose waves share a ring of `nslots` big LDS stages
// (two stages take the whole 160 KB: ~140 rows each at w = 128, so molecules and graphs of a few hundred nodes fit).
// The workgroup walks a contiguous run of node tiles (whole graphs).  Per stage every wave fires its share of the
// LDS-DMA of the rows, node records, normalisers and the stage's CSR slice (global_load_lds, no VGPRs); stages retire
// in order behind a COUNTED vmcnt wait (VM operations retire in issue order, so "at most n younger operations
// outstanding" proves the stage has landed while the next stage's DMA and the previous stage's output stores stay in
// flight) and two barriers.  At the BASELINE sizes a CU's whole share of the input fits its ring, so all reads are in
// flight from the first microsecond; bigger batches cycle the ring.  A tile that does not fit a stage (one very large
// graph) is reduced straight from global memory.  Forms measured and dropped this round: a short-lived
// workgroup per tile group (round 1's default), one ring per wave.
static constexpr int RING_MAX_SLOTS = 4;

// Diagnostic build: wave 0 of every workgroup logs wall-clock stamps of its stage events (100 MHz ticks)
#ifdef GNNB_PROBE
#define RING_EV(code)                                                                                        \
    do {                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 2048 && pev < 31) {                                             \
            g_probe[blockIdx.x * 64 + 2 + 2 * pev] = wall_clock64();                                         \
            g_probe[blockIdx.x * 64 + 3 + 2 * pev] = (unsigned long long)(code);                             \
            pev++;                                                                                           \
            g_probe[blockIdx.x * 64 + 1] = pev;                                                              \
        }                                                                                                    \
    } while (0)
#else
#define RING_EV(code) do { } while (0)
#endif

template <int MODE, int VEC, bool NT>
__global__ __launch_bounds__(1024) void k_aggregate_ring(
    const float *__restrict__ x, const float *__restrict__ selfq, float *__restrict__ out,
    const int4 *__restrict__ node_rec, const int32_t *__restrict__ col, const float *__restrict__ dinv,
    const int32_t *__restrict__ tile_first, const int32_t *__restrict__ tile_edge, int num_tiles, int N, int E, int w,
    int glog2, int cap, int ecap, int nslots, int slot_bytes, int slack, float eps, int tile_lo,
    const float4 *__restrict__ gcoef, const int4 *__restrict__ cut, int tile_rows)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool HASQ = MODE == GNNB_AGG_PNA, HASREC = MODE != GNNB_AGG_COPY, HASDINV = MODE == GNNB_AGG_GCN;
    // GCN, float4 rows: the coefficients dinv_i dinv_j of the four inline neighbours come precomputed (k_gcn_coef: 0 for the
    // unused slots), one 16-B LDS read per row instead of four normaliser reads, four products and four selects
    constexpr bool HASGC = MODE == GNNB_AGG_GCN && VEC == 4;
    constexpr int KOUT = AggOut<MODE>::K;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    // the workgroup's waves share the ring: wave sw of sn issues 1/sn of a stage's DMA and reduces 1/sn of its rows
    const int sw = wave, sn = nw;
    // (tile_lo > 0: only the tiles of the caller's large segment, see gnnb_workspace_set_large_segment)
    // The workgroup's rows.  Without `cut`: a run of whole node tiles (= whole graphs), equal tile counts -- rows +- one graph,
    // 288 +- 8 % at BASELINE config 2, and the kernel ends with its slowest workgroup.  With `cut` (graph prep's row-balanced
    // ranges, round 4): rows [r_lo, r_hi) = an exact 1 / grid share; the graphs those rows belong to are staged WHOLE --
    // the boundary graph by both neighbours (~3 % more reads) -- and every workgroup reduces and stores only its own rows.
    int t0, t1, r_lo = 0, r_hi = N, gs_row = 0, gs_edge = 0;
    const bool use_cut = cut != nullptr;
    if (use_cut) {
        const int4 c0 = cut[blockIdx.x], c1 = cut[blockIdx.x + 1];
        // (clamped: the table of a malformed -- flagged -- batch may hold stale entries; it must still stay in range)
        r_lo = min(max(c0.x, 0), N);
        r_hi = min(max(c1.x, r_lo), N);
        gs_row = min(max(c0.y, 0), r_lo); // first row of the graph that owns r_lo
        gs_edge = min(max(c0.z, 0), E);
        t0 = min(gs_row / tile_rows, num_tiles);                          // (tile_first[t0] <= gs_row: never read as a stage start)
        t1 = min((r_hi + tile_rows - 1) / tile_rows, num_tiles);          // tile_first[t1] = a graph start >= r_hi
        if (r_hi <= r_lo)
            return;
    } else {
        run_cuts(blockIdx.x, gridDim.x, (unsigned)(num_tiles - tile_lo), t0, t1); // (32-bit: gnnb_device.h)
        t0 += tile_lo;
        t1 += tile_lo;
    }
    if (t0 >= t1)
        return; // (workgroup-uniform)
#ifdef GNNB_PROBE
    int pev = 0;
    if (threadIdx.x == 0 && blockIdx.x < 2048)
        g_probe[blockIdx.x * 64] = wall_clock64();
#endif
    char *wbase = smem;
    const int off_q = cap * w * 4;
    const int off_rec = off_q + (HASQ ? cap * w * 4 : 0);
    const int off_dinv = off_rec + cap * 32;
    const int off_gc = off_dinv + ((cap * 4 + 15) & ~15);
    const int off_col = off_gc + (HASGC ? cap * 16 : 0); // the stage's CSR slice (rows of degree > 4 read it): ecap entries
    const int nvec = w / VEC;
    const int G = 1 << glog2;       // lanes per destination row
    const int groups = 64 >> glog2; // rows a wave reduces at once
    const int grp = lane >> glog2;
    const int gl = lane & (G - 1);
    const int fiter = (nvec + G - 1) / G;

    // window of the tile table in a register: lane l holds tile_first[wb + l]
    // (clamped to N: the table of a malformed batch may hold stale entries; a flagged batch must still stay in range)
    int wb = t0;
    int tfv = min(tile_first[min(wb + lane, num_tiles)], N);
    int tev = HASREC ? min(tile_edge[min(wb + lane, num_tiles)], E) : 0;
    const int tf_end = min(tile_first[t1], N); // end of this ring's node range
    int ts = t0; // next tile to plan
    bool first_cut = use_cut; // (row-balanced ranges: the first stage starts at the cut's graph, not at tile_first[t0])

    int f_nb[RING_MAX_SLOTS], f_rows[RING_MAX_SLOTS], f_mark[RING_MAX_SLOTS], f_e0[RING_MAX_SLOTS];
    int nfifo = 0, vm = 0, issue_slot = 0, head_slot = 0;

    auto issue = [&](int slot, int nb_, int rows_, int e0_, int ne_) -> int {
        char *sb = wbase + (size_t)slot * slot_bytes;
        int ops = 0;
        const int bytes = rows_ * w * 4;
        {
            const char *gx = reinterpret_cast<const char *>(x + (size_t)nb_ * w);
            if (VEC == 4) {
                for (int c = sw * 1024; c < bytes; c += sn * 1024, ops++)
                    if (c + lane * 16 < bytes) {
#ifdef GNNB_AGG_NT_LOAD // (development A/B, round 6: the feature rows' LDS-DMA with the non-temporal policy -- every row is read once)
                        const uint32_t a_ = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_vptr)(sb + c));
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(a_), "v"(gx + c + lane * 16) : "memory");
#else
                        dma16_to_lds_u(gx + c + lane * 16, sb + c);
#endif
                    }
            } else {
                for (int c = sw * 64; c < rows_ * w; c += sn * 64, ops++)
                    if (c + lane < rows_ * w)
                        dma4_to_lds_u(gx + (size_t)(c + lane) * 4, sb + (size_t)c * 4);
            }
        }
        if (HASQ && selfq != nullptr) {
            const char *gq = reinterpret_cast<const char *>(selfq + (size_t)nb_ * w);
            if (VEC == 4) {
                for (int c = sw * 1024; c < bytes; c += sn * 1024, ops++)
                    if (c + lane * 16 < bytes)
                        dma16_to_lds_u(gq + c + lane * 16, sb + off_q + c);
            } else {
                for (int c = sw * 64; c < rows_ * w; c += sn * 64, ops++)
                    if (c + lane < rows_ * w)
                        dma4_to_lds_u(gq + (size_t)(c + lane) * 4, sb + off_q + (size_t)c * 4);
            }
        }
        if (HASREC) {
            const char *grec = reinterpret_cast<const char *>(node_rec + 2 * (size_t)nb_);
            const int rbytes = rows_ * 32;
            for (int c = sw * 1024; c < rbytes; c += sn * 1024, ops++)
                if (c + lane * 16 < rbytes)
                    dma16_to_lds_u(grec + c + lane * 16, sb + off_rec + c);
        }
        if (HASDINV) {
            for (int c = sw * 64; c < rows_; c += sn * 64, ops++)
                if (c + lane < rows_)
                    dma4_to_lds_u(dinv + nb_ + c + lane, sb + off_dinv + (size_t)c * 4);
        }
        if (HASGC) {
            const char *gg = reinterpret_cast<const char *>(gcoef + nb_);
            const int gbytes = rows_ * 16;
            for (int c = sw * 1024; c < gbytes; c += sn * 1024, ops++)
                if (c + lane * 16 < gbytes)
                    dma16_to_lds_u(gg + c + lane * 16, sb + off_gc + c);
        }
        if (HASREC) { // (a tracked global read of col inside the reduction would drain this wave's whole pipeline)
            for (int c = sw * 64; c < ne_; c += sn * 64, ops++)
                if (c + lane < ne_)
                    dma4_to_lds_u(col + e0_ + c + lane, sb + off_col + (size_t)c * 4);
        }
        return ops;
    };

    // reduce one landed stage; returns the number of store instructions the wave issued
    auto compute = [&](int slot, int nb_, int rows_, int e0_) -> int {
        // (this workgroup's rows of the stage: all of them, or -- row-balanced ranges -- the part inside [r_lo, r_hi))
        const int clo = max(r_lo - nb_, 0), chi = min(r_hi - nb_, rows_);
        const char *sb = wbase + (size_t)slot * slot_bytes;
        const float *sx = reinterpret_cast<const float *>(sb);
        const float *sq = reinterpret_cast<const float *>(sb + off_q);
        const int4 *srec = reinterpret_cast<const int4 *>(sb + off_rec);
        const float *sdinv = reinterpret_cast<const float *>(sb + off_dinv);
        const int32_t *scol = reinterpret_cast<const int32_t *>(sb + off_col) - e0_; // indexed by the CSR slot itself
        int nst = 0;
        if (VEC == 4 && (MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_SUM || MODE == GNNB_AGG_MEAN || MODE == GNNB_AGG_SIMPLE) && nvec <= G) {
            // lean path (LdsRowF): one column pass (a lane group covers the row), two rows per lane group in flight
            const int w4 = w * 4;
            const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_vptr)sb;
            const uint32_t sx_lane = lds0 + (uint32_t)gl * 16u;
            const uint32_t sxo = sx_lane - (uint32_t)__mul24(nb_, w4);
            const uint32_t sdo = lds0 + (uint32_t)off_dinv - (uint32_t)nb_ * 4u;
            const uint32_t scol_o = lds0 + (uint32_t)off_col - (uint32_t)e0_ * 4u;
            const float4 *sgc = reinterpret_cast<const float4 *>(sb + off_gc);
            char *obase = reinterpret_cast<char *>(out + (size_t)nb_ * w) + gl * 16;
            const bool lane_on = gl < nvec;
            for (int rb = clo + sw * 2 * groups; rb < chi; rb += sn * 2 * groups) {
                const bool has_b = rb + groups < chi;
                LdsRowF<MODE == GNNB_AGG_PNA || MODE == GNNB_AGG_LG || MODE == GNNB_AGG_COPY ? GNNB_AGG_SUM : MODE, NT> A, B;
                A.begin(lane_on && rb + grp < chi, rb + grp, srec, sxo, sx_lane, sdinv, sgc, w4);
                if (has_b)
                    B.begin(lane_on && rb + groups + grp < chi, rb + groups + grp, srec, sxo, sx_lane, sdinv, sgc, w4);
                A.finish(sxo, sdo, scol_o, w4, obase, eps);
                if (has_b)
                    B.finish(sxo, sdo, scol_o, w4, obase, eps);
                nst += has_b ? 2 : 1;
            }
            return nst;
        }
        for (int rb = clo + sw * 2 * groups; rb < chi; rb += sn * 2 * groups) {
            const bool has_b = rb + groups < chi; // wave-uniform: the second row's store exists or not for the whole wave
            for (int f = gl; f < nvec; f += G) {
                const int fo = f * VEC;
                LdsRow<MODE, VEC, true, NT> A, B;
                A.begin(rb + grp < chi, nb_, rb + grp, sx, sq, srec, sdinv, selfq, w, fo);
                if (has_b)
                    B.begin(rb + groups + grp < chi, nb_, rb + groups + grp, sx, sq, srec, sdinv, selfq, w, fo);
                A.finish(nb_, sx, srec, sdinv, scol, out, w, fo, eps);
                if (has_b)
                    B.finish(nb_, sx, srec, sdinv, scol, out, w, fo, eps);
            }
            nst += KOUT * fiter * (has_b ? 2 : 1);
        }
        return nst;
    };

    for (;;) {
        // ---- fill the ring: plan greedy stages of whole tiles and fire their DMA
        while (nfifo < nslots && ts < t1) {
            int rel = __builtin_amdgcn_readfirstlane(ts - wb);
            if (rel >= 32 && wb + 63 < t1) { // slide the window (a tracked load: drain first so the counts stay exact)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wb = ts;
                tfv = min(tile_first[min(wb + lane, num_tiles)], N);
                tev = HASREC ? min(tile_edge[min(wb + lane, num_tiles)], E) : 0;
                rel = 0;
            }
            const int nb_ = first_cut ? gs_row : __builtin_amdgcn_readlane(tfv, rel);
            const int e0_ = first_cut ? gs_edge : __builtin_amdgcn_readlane(tev, rel);
            first_cut = false;
            // a stage = a run of whole tiles whose rows AND CSR slice fit a slot.  The remaining rows are cut into
            // EQUAL stages (a greedy cut leaves a tiny last stage, and every stage costs a memory latency when the
            // ring is shallower than the range): aim at remaining / ceil(remaining / cap) rows, + half a tile
            const int rem = max(tf_end - nb_, 1);
            const int nrem = (rem + cap - 1) / cap;
            const int cap_eff = min(cap, (rem + nrem - 1) / nrem + slack);
            unsigned long long m = __ballot(lane > rel && wb + lane <= t1 && tfv - nb_ <= cap_eff && tev - e0_ <= ecap &&
                                            tev >= e0_);
            if (m == 0 && cap_eff < cap)
                m = __ballot(lane > rel && wb + lane <= t1 && tfv - nb_ <= cap && tev - e0_ <= ecap && tev >= e0_);
            int te;
            bool big = false;
            if (m == 0) { // the next tile alone exceeds a stage
                te = ts + 1;
                big = true;
            } else {
                te = wb + 63 - __builtin_clzll(m);
            }
            const int rows_ = __builtin_amdgcn_readlane(tfv, __builtin_amdgcn_readfirstlane(te - wb)) - nb_;
            const int ne_ = big ? 0 : __builtin_amdgcn_readlane(tev, __builtin_amdgcn_readfirstlane(te - wb)) - e0_;
            ts = te;
            if (rows_ <= 0)
                continue;
            if (big) {
                for (int r = max(r_lo - nb_, 0) + sw * groups + grp; r < min(r_hi - nb_, rows_); r += sn * groups)
                    for (int f = gl; f < nvec; f += G)
                        agg_row_direct<MODE, VEC, NT>(nb_ + r, x, selfq, out, node_rec, col, dinv, w, f * VEC, eps);
                continue;
            }
            vm += issue(issue_slot, nb_, rows_, e0_, ne_);
            RING_EV(1000000 + rows_); // stage issued
#pragma unroll
            for (int i = 0; i < RING_MAX_SLOTS; i++)
                if (i == nfifo) {
                    f_nb[i] = nb_;
                    f_rows[i] = rows_;
                    f_mark[i] = vm;
                    f_e0[i] = e0_;
                }
            nfifo++;
            issue_slot = issue_slot + 1 == nslots ? 0 : issue_slot + 1;
        }
        if (nfifo == 0)
            break;
        // ---- retire the oldest stage: everything issued after its DMA may stay in flight
        RING_EV(2000000 + f_rows[0]); // waiting for the oldest stage
        vmcnt_wait_n(min(vm - f_mark[0], 63));
        asm volatile("s_barrier" ::: "memory"); // every wave's share of the stage has landed
        RING_EV(3000000 + f_rows[0]); // landed
        vm += compute(head_slot, f_nb[0], f_rows[0], f_e0[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // its LDS reads are done before the slot is refilled
        asm volatile("s_barrier" ::: "memory"); // ... by every wave
        RING_EV(4000000 + f_rows[0]); // reduced, stores issued
#pragma unroll
        for (int i = 0; i + 1 < RING_MAX_SLOTS; i++) {
            f_nb[i] = f_nb[i + 1];
            f_rows[i] = f_rows[i + 1];
            f_mark[i] = f_mark[i + 1];
            f_e0[i] = f_e0[i + 1];
        }
        nfifo--;
        head_slot = head_slot + 1 == nslots ? 0 : head_slot + 1;
    }
}

template <int MODE, int VEC>
static hipError_t launch_aggregate_ring_t(const BatchTables &t, const float *x, const float *selfq,
                                          float *out, int w, float eps, hipStream_t s)
{
    const Options &o = options();
    const int tile_lo = std::min(std::max(t.tile_lo, 0), t.num_tiles);
    if (t.num_tiles - tile_lo <= 0)
        return hipSuccess;
    const int nvec = w / VEC;
    int glog2 = 0;
    while ((1 << glog2) < nvec && glog2 < 6)
        glog2++;
    const int num_cus = device_cu_count();
    // per staged row: the row itself (PNA: p and q), its 32-B record, its normaliser, and 4 CSR entries (a stage
    // whose CSR slice is longer than 4 per row -- multigraphs, hubs -- is cut shorter by the planner)
    constexpr int ECAP_PER_ROW = 4;
    const size_t per_row = (size_t)w * 4 * (MODE == GNNB_AGG_PNA ? 2 : 1) + (MODE != GNNB_AGG_COPY ? 32 + 4 + 4 * ECAP_PER_ROW : 0) +
                           (MODE == GNNB_AGG_GCN && VEC == 4 ? 16 : 0);
    const int wgs = std::max((int)o.agg_ring_wg_per_cu, 1);
    const size_t budget = (size_t)(o.agg_lds_kb > 0 ? std::min(std::max((int)o.agg_lds_kb, 8), 158) : 158 / wgs) * 1024;
    int ns = std::min(std::max((int)o.agg_ring_slots, 1), RING_MAX_SLOTS);
    int nw = o.agg_ring_waves;
    // one ring per workgroup: stages as large as the budget allows
    if (nw <= 0)
        nw = 16; // (measured: 16 waves issue a stage's DMA and drain its stores faster than 8)
    nw = std::min(std::max(nw, 1), 16);
    int cap = (int)((budget / ns) / per_row);
    cap = std::min(std::max(cap, 1), 4096);
    const int slot_bytes = (int)((((size_t)cap * per_row) + 31) & ~(size_t)15); // (+ 16: the normalisers are padded to 16 B)
    const size_t lds = (size_t)ns * slot_bytes;
    // persistent: `wgs` workgroups per CU; fewer when the batch has fewer tiles than rings
    int grid = num_cus * wgs;
    grid = std::min(grid, t.num_tiles - tile_lo);
    if (grid < 1)
        grid = 1;
    // graph prep's row-balanced ranges, when they were made for exactly this grid
    const int4 *cut = (o.agg_balance && tile_lo == 0 && t.agg_cut && t.agg_cut_n == grid) ? t.agg_cut : nullptr;
    auto launch = [&](auto kern) -> hipError_t {
        {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
            if (e != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, s, x, selfq, out, t.node_rec, t.col, t.dinv,
                           t.tile_first, t.tile_edge, t.num_tiles, t.num_nodes, t.num_edges, w, glog2, cap,
                           cap * ECAP_PER_ROW, ns, slot_bytes, std::max(t.tile_rows / 2, 1) + 2, eps, tile_lo, t.gcoef, cut,
                           std::max(t.tile_rows, 1));
        return hipGetLastError();
    };
    if (o.agg_nt_store)
        return launch(k_aggregate_ring<MODE, VEC, true>);
    return launch(k_aggregate_ring<MODE, VEC, false>);
}

// -------------------------------------------------------------------------------------
// GINE aggregate (reference gine_conv_agg + the self term of gine_conv, gnn_builder_lib.h:1555-1742):
//   out_i = (1 + eps) x_i + sum_{j -> i} relu(x_j + p_e),   p_e = W_e e_ij + b_e  (projected by the GEMM kernel,
// [E, w] rows in COO order; the CSR slot's COO row comes from the edge-index table graph prep writes,
// compute_neighbor_and_edge_index_tables :1126-1166).  One lane group per destination row, rows and edge terms
// gathered straight from L2: GINE is not on a BASELINE workload, this is the plain form.
template <int VEC>
__global__ __launch_bounds__(WG) void k_aggregate_edges(const float *__restrict__ x, const float *__restrict__ eterm,
                                                        float *__restrict__ out, const int4 *__restrict__ node_rec,
                                                        const int32_t *__restrict__ col, const int32_t *__restrict__ eid,
                                                        int N, int w, int glog2, float eps)
{
    typedef Vf<VEC> V;
    const int G = 1 << glog2, groups = WG >> glog2;
    const int grp = threadIdx.x >> glog2, gl = threadIdx.x & (G - 1);
    const int node = blockIdx.x * groups + grp;
    if (node >= N)
        return;
    const int4 r0 = node_rec[2 * (size_t)node];
    const int rp0 = r0.x, deg = r0.y;
    const int nvec = w / VEC;
    for (int f = gl; f < nvec; f += G) {
        const int fo = f * VEC;
        V acc = V::splat(0.0f);
        for (int k = rp0; k < rp0 + deg; k++) {
            const V xj = V::load(x + (size_t)col[k] * w + fo);
            const V pe = V::load(eterm + (size_t)eid[k] * w + fo);
            acc = vadd(acc, vmax(vadd(xj, pe), V::splat(0.0f))); // merge_sum_1d, activation_relu, sum_incremental
        }
        const V xi = V::load(x + (size_t)node * w + fo);
        vadd(acc, vmul(xi, V::splat(1.0f + eps))).store(out + (size_t)node * w + fo);
    }
}

hipError_t launch_aggregate_edges(const BatchTables &t, const float *x, const float *eterm, float *out, int width,
                                  float eps, hipStream_t s)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    const bool v4 = (width % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)out & 15) == 0) &&
                    (((uintptr_t)eterm & 15) == 0);
    const int nvec = v4 ? width / 4 : width;
    int glog2 = 0;
    while ((1 << glog2) < nvec && glog2 < 6)
        glog2++;
    const int groups = WG >> glog2;
    const int grid = (t.num_nodes + groups - 1) / groups;
    if (v4)
        hipLaunchKernelGGL(k_aggregate_edges<4>, dim3(grid), dim3(WG), 0, s, x, eterm, out, t.node_rec, t.col, t.eid,
                           t.num_nodes, width, glog2, eps);
    else
        hipLaunchKernelGGL(k_aggregate_edges<1>, dim3(grid), dim3(WG), 0, s, x, eterm, out, t.node_rec, t.col, t.eid,
                           t.num_nodes, width, glog2, eps);
    return hipGetLastError();
}

// GCN coefficients of the inline neighbours: gcoef[v] = dinv_v * {dinv_j0 .. dinv_j3}, 0 for the slots past the degree.
// A derived table like dinv itself; written once per prepared batch in front of the first layer-wise GCN aggregate
// (the LDS-resident stack kernels compute the same products in their P0 phase and do not use it).
__global__ __launch_bounds__(WG) void k_gcn_coef(const int4 *__restrict__ node_rec, const float *__restrict__ dinv, int N,
                                                 float4 *__restrict__ gcoef)
{
    const int v = blockIdx.x * WG + threadIdx.x;
    if (v >= N)
        return;
    const int4 r0 = node_rec[2 * (size_t)v], r1 = node_rec[2 * (size_t)v + 1];
    const int deg = r0.y;
    const float di = dinv[v];
    gcoef[v] = make_float4(deg > 0 ? di * dinv[r0.z] : 0.0f, deg > 1 ? di * dinv[r0.w] : 0.0f, deg > 2 ? di * dinv[r1.x] : 0.0f,
                           deg > 3 ? di * dinv[r1.y] : 0.0f);
}

hipError_t launch_gcn_coef(const BatchTables &t, hipStream_t s)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_gcn_coef, dim3((t.num_nodes + WG - 1) / WG), dim3(WG), 0, s, t.node_rec, t.dinv, t.num_nodes, t.gcoef);
    return hipGetLastError();
}

int aggregate_ring_grid() { return device_cu_count() * std::max((int)options().agg_ring_wg_per_cu, 1); }

hipError_t launch_aggregate(const BatchTables &t, int kind, const float *x, const float *selfq,
                            float *out, int width, float eps, hipStream_t s)
{
    // agg_form: 0 = the ring (default), 1 = the register-gather form (k_aggregate_rg.hip) wherever it exists, 2 = that form for PNA
    // only.  Measured (round 5, same box, HBM regime, us per launch ring / register gather): GCN w = 128 at BASELINE config 2
    // 15.0-15.3 / 14.7-15.3 (a wash); PNA at config 4 72.7 / 65.0 solo -- but the config-4 STEP is 1.2 % slower with it (844.5
    // against 835 us: its thousands of short waves take issue slots from the GEMMs of the other batches in flight); MEAN
    // w = 256 at config 5 80-81 / 81; SUM at config 3 22.2 / 27.4
    if (options().agg_form == 1 || (options().agg_form == 2 && kind == GNNB_AGG_PNA)) {
        const hipError_t e = launch_aggregate_rg(t, kind, x, selfq, out, width, eps, s);
        if (e != hipErrorNotSupported)
            return e;
    }
    const bool v4 = (width % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)out & 15) == 0) &&
                    (selfq == nullptr || ((uintptr_t)selfq & 15) == 0);
#define GNNB_AGG_CASE(K)                                                                         \
    case K:                                                                                      \
        return v4 ? launch_aggregate_ring_t<K, 4>(t, x, selfq, out, width, eps, s)               \
                  : launch_aggregate_ring_t<K, 1>(t, x, selfq, out, width, eps, s);
    switch (kind) {
        GNNB_AGG_CASE(GNNB_AGG_GCN)
        GNNB_AGG_CASE(GNNB_AGG_SUM)
        GNNB_AGG_CASE(GNNB_AGG_MEAN)
        GNNB_AGG_CASE(GNNB_AGG_PNA)
        GNNB_AGG_CASE(GNNB_AGG_LG)
        GNNB_AGG_CASE(GNNB_AGG_SIMPLE)
        GNNB_AGG_CASE(GNNB_AGG_COPY)
    default:
        return hipErrorInvalidValue;
    }
#undef GNNB_AGG_CASE
}


} // namespace gnnb
