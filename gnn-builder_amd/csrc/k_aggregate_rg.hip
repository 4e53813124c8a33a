// k_aggregate_rg.hip -- gather -> segmented reduce per destination row, REGISTER-GATHER form (round 5)
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
//
// Reference: gcn_conv_agg / gin_conv_agg / sage_conv_agg / pna_conv_agg / simple_conv (gnn_builder_lib.h:1213-1289, :1389-1437,
// :2161-2209, :1750-1834, :2501-2634): `for node: for neighbour: for feature` with one owner per output row.
//
// The ring form (k_aggregate.hip) stages whole graphs in LDS behind workgroup barriers: DMA -> wait -> reduce -> store, in
// series per CU, two barriers per stage -- it takes the same 15.5 us at BASELINE config 2 whether its bytes come from the
// Infinity Cache or from HBM: it is bound by its own per-CU pipeline, not by bandwidth (round-4 review).  This form has
// NO LDS and NO barrier.  A lane group of w / 4 lanes (float4 per lane: 32 lanes at w = 128, i.e. two rows per wave
// instruction) owns a destination row; every wave owns a contiguous run of rows (a molecule's worth), cut into batches of
// R row-instructions.  Per batch a wave
//   1. holds the node records of the batch (prefetched one batch ahead: {row start, degree, first four sources} -- 32 B a
//      row, every lane of a group reads the same address: one request),
//   2. fires the batch's whole gather -- own row + up to four source rows per destination, global_load_dwordx4, 5 R loads in
//      flight per lane; sources are rows of the same molecule (<= 15 KB away): served by the CU's vector L1 / the XCD's L2,
//      each row leaves HBM once --, then the next batch's records,
//   3. reduces in CSR order (sources first, self term last: the reference's order, and the ring form's bits) as the loads
//      retire in order (counted vmcnt by the compiler) and stores non-temporally, 16 B a lane, whole rows.
// Sixteen or more such waves per CU overlap each other's phases without any hand-written schedule.  Workgroups are mapped
// XCD-aware: workgroup b runs on XCD b mod 8, and the rows are dealt so that every XCD owns ONE contiguous eighth of the
// batch -- a row and the rows that gather it meet in one L2.
// Rows of degree > 4 read the rest of their CSR row from `col` (dependent loads; rare for molecules).
// HBM traffic = 4 w N read + 4 w N k_out write + 32 N records (+ 20 N GCN coefficients): the ring form's.
#include "gnnb_device.h"

namespace gnnb {

template <int MODE>
struct RgTraits {
    static constexpr bool self = MODE == GNNB_AGG_GCN || MODE == GNNB_AGG_SUM; // reads the row itself
    static constexpr int kout = MODE == GNNB_AGG_PNA ? 4 : 1;
};

typedef float rg_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rg_f4 rg_ld(const float *p) { return *reinterpret_cast<const rg_f4 *>(p); }
template <bool NT>
__device__ __forceinline__ void rg_st(float *p, rg_f4 v)
{
    if (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<rg_f4 *>(p));
    else
        *reinterpret_cast<rg_f4 *>(p) = v;
}

// MODE: GNNB_AGG_GCN | SUM | MEAN | SIMPLE | PNA (PNA: x = p, selfq = q or nullptr; out [N, 4 w] = max | min | mean | std).
// GLOG2: log2(lanes per row), w = 4 << GLOG2.  R: row-instructions per batch.  PRED: sources past the degree are not
// loaded (exec-masked) instead of aliasing the row itself with coefficient 0.  MINB: workgroups per CU the register
// allocation must leave room for.
template <int MODE, int GLOG2, int R, int FLAGS, int MINB>
__global__ __launch_bounds__(256, MINB) void k_aggregate_rg(const float *__restrict__ x, const float *__restrict__ selfq,
                                                            float *__restrict__ out, const int4 *__restrict__ node_rec,
                                                            const int32_t *__restrict__ col, const float *__restrict__ dinv,
                                                            const float4 *__restrict__ gcoef, int N, int rpw, float eps)
{
    constexpr int G = 1 << GLOG2, RPI = 64 >> GLOG2, W = 4 * G;
    // FLAGS: 1 = sources past the degree are not loaded, 2 = the run is pre-touched line by line (measured: slower),
    // ablations (diagnostics, wrong results): 4 = no source loads (the row itself stands in), 8 = no stores
    constexpr bool PRED = (FLAGS & 1) != 0, PF = (FLAGS & 2) != 0, NO_NBR = (FLAGS & 4) != 0, NO_ST = (FLAGS & 8) != 0, NT = true;
    constexpr bool GCN = MODE == GNNB_AGG_GCN, PNA = MODE == GNNB_AGG_PNA, SELF = RgTraits<MODE>::self;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane >> GLOG2, gl = lane & (G - 1);
    // XCD-aware: consecutive workgroup ids land on consecutive XCDs; XCD c gets the contiguous eighth c of the rows
    const int nb8 = gridDim.x >> 3;
    const int vb = (gridDim.x & 7) == 0 ? ((int)blockIdx.x & 7) * nb8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int r0 = (vb * 4 + wave) * rpw, r1 = min(r0 + rpw, N);
    if (r0 >= r1)
        return; // (wave-uniform)
    const float *const xl = x + gl * 4;
    const bool hasq = PNA && selfq != nullptr;

    // Software pipeline over the wave's batches, ordered so that NO wait ever covers a store (gfx950 counts loads and stores
    // in one vmcnt, retired in issue order): per iteration  reduce(b) -> issue gather(b+1) -> issue records(b+2) -> store(b).
    // A wait for gather(b+1) leaves records(b+2) and the stores of b outstanding; a wait for records(b+2) leaves the stores
    // outstanding, which are UNCONDITIONAL so that the compiler can count them: lane groups past the run re-compute the
    // run's last row from the same loads and store the same bytes to the same place.
    struct Batch {
        rg_f4 xi[R], nv[R][4], gc[R];
        float di[R];
        int row[R], rp0[R], deg[R];
    };
    int4 ra[R], rc_[R]; // records: {row start, degree, j0, j1}{j2, j3, -, -}
    auto load_recs = [&](int rb) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int row = min(rb + k * RPI + grp, r1 - 1); // (past the run: its last row again)
            ra[k] = node_rec[2 * (size_t)row];
            rc_[k] = node_rec[2 * (size_t)row + 1];
        }
    };
    auto gather = [&](int rb, Batch &B) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int row = min(rb + k * RPI + grp, r1 - 1);
            B.row[k] = row;
            B.rp0[k] = ra[k].x;
            B.deg[k] = ra[k].y;
            const int j[4] = {ra[k].z, ra[k].w, rc_[k].x, rc_[k].y}; // batch-global ids; unused slots alias the row itself
            if (SELF)
                B.xi[k] = rg_ld(xl + (size_t)row * W);
            else if (PNA)
                B.xi[k] = hasq ? rg_ld(selfq + (size_t)row * W + gl * 4) : rg_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (NO_NBR) {
                    B.nv[k][q] = SELF ? B.xi[k] : rg_f4{1.f, 1.f, 1.f, 1.f};
                } else if (PRED) {
                    B.nv[k][q] = rg_f4{0.f, 0.f, 0.f, 0.f};
                    if (B.deg[k] > q)
                        B.nv[k][q] = rg_ld(xl + (size_t)j[q] * W);
                } else {
                    B.nv[k][q] = rg_ld(xl + (size_t)j[q] * W);
                }
            }
            if (GCN) {
                const float4 g4 = gcoef[row]; // dinv_i dinv_j of the inline sources, 0 past the degree (k_gcn_coef)
                B.gc[k] = rg_f4{g4.x, g4.y, g4.z, g4.w};
                B.di[k] = dinv[row];
            }
        }
    };
    constexpr int KO = RgTraits<MODE>::kout;
    // reduce row-instruction k of a landed batch in CSR order (sources first, self term last) -> res[0 .. KO)
    auto reduce = [&](const Batch &B, int k, rg_f4 (&res)[KO]) {
        const int d = B.deg[k];
        if (!PNA) {
            rg_f4 c;
            if (GCN)
                c = B.gc[k];
            else
                c = rg_f4{d > 0 ? 1.f : 0.f, d > 1 ? 1.f : 0.f, d > 2 ? 1.f : 0.f, d > 3 ? 1.f : 0.f};
            // (the first term initialises the sum: 0 + v c, as the ring form and the reference start from zero)
            rg_f4 acc = B.nv[k][0] * c.x;
            acc = acc + B.nv[k][1] * c.y;
            acc = acc + B.nv[k][2] * c.z;
            acc = acc + B.nv[k][3] * c.w;
            if (d > 4) { // the rest of the CSR row (dependent loads: rare for molecules)
                for (int e = B.rp0[k] + 4; e < B.rp0[k] + d; e++) {
                    const int jj = col[e];
                    const rg_f4 v = rg_ld(xl + (size_t)jj * W);
                    const float cj = GCN ? B.di[k] * dinv[jj] : 1.0f;
                    acc = acc + v * cj;
                }
            }
            if (GCN)
                res[0] = acc + B.xi[k] * (B.di[k] * B.di[k]);
            else if (MODE == GNNB_AGG_SUM)
                res[0] = acc + B.xi[k] * (1.0f + eps);
            else if (MODE == GNNB_AGG_MEAN)
                res[0] = d > 0 ? acc * (1.0f / (float)d) : acc;
            else
                res[0] = acc;
        } else {
            // PNA: h_j = q_i + p_j; max | min | mean | std over j (PyG's std: SURVEY finding 5)
            rg_f4 vmx = rg_f4{0.f, 0.f, 0.f, 0.f}, vmn = vmx, s1 = vmx, s2 = vmx;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (d > q) {
                    const rg_f4 h = B.xi[k] + B.nv[k][q];
                    if (q == 0) {
                        vmx = h;
                        vmn = h;
                    } else {
                        vmx = __builtin_elementwise_max(vmx, h);
                        vmn = __builtin_elementwise_min(vmn, h);
                    }
                    s1 = s1 + h;
                    s2 = s2 + h * h;
                }
            }
            if (d > 4) {
                for (int e = B.rp0[k] + 4; e < B.rp0[k] + d; e++) {
                    const rg_f4 h = B.xi[k] + rg_ld(xl + (size_t)col[e] * W);
                    vmx = __builtin_elementwise_max(vmx, h);
                    vmn = __builtin_elementwise_min(vmn, h);
                    s1 = s1 + h;
                    s2 = s2 + h * h;
                }
            }
            rg_f4 mean = rg_f4{0.f, 0.f, 0.f, 0.f}, sd = mean;
            if (d > 0) {
                const float dn = (float)d;
                mean = s1 / dn;
                const rg_f4 m2 = s2 / dn;
                sd = rg_f4{pyg_std1(m2.x, mean.x), pyg_std1(m2.y, mean.y), pyg_std1(m2.z, mean.z), pyg_std1(m2.w, mean.w)};
            }
            res[0] = vmx;
            res[KO > 1 ? 1 : 0] = vmn;
            res[KO > 2 ? 2 : 0] = mean;
            res[KO > 3 ? 3 : 0] = sd;
        }
    };

    // PF: the wave's whole run of rows is requested at once, one dword per 128-B line and lane (up to 256 lines = 32 KB with
    // four instructions and four registers): every row of the run is on its way from HBM from the first cycle, and the
    // batches' own gathers find it in the XCD's L2
    float pf[4] = {0.f, 0.f, 0.f, 0.f};
    if (PF) {
        const int lines = (r1 - r0) * (W / 32);
        const float *xr = x + (size_t)r0 * W;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (lane + 64 * i < lines)
                pf[i] = xr[(size_t)(lane + 64 * i) * 32];
    }
    Batch B;
    load_recs(r0);
    gather(r0, B);
    load_recs(r0 + R * RPI);
    for (int rb = r0; rb < r1; rb += R * RPI) {
        rg_f4 res[R][KO];
        int orow[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            reduce(B, k, res[k]);
            orow[k] = B.row[k];
        }
        if (rb + R * RPI < r1) { // (wave-uniform)
            gather(rb + R * RPI, B);
            load_recs(rb + 2 * R * RPI);
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            float *o = out + (size_t)orow[k] * (KO * W) + gl * 4;
#pragma unroll
            for (int c = 0; c < KO; c++)
                if (!NO_ST || res[k][c].x == 12345.678f) // (ablation: the results stay live, nothing is stored)
                    rg_st<NT>(o + c * W, res[k][c]);
        }
    }
    if (PF) { // (keeps the touches alive; never true for finite data: |sum| of four finite floats is not +inf)
        const float t = (pf[0] + pf[1]) + (pf[2] + pf[3]);
        if (__builtin_isinf(t) && t > 0.f && eps == -12345.f)
            out[0] = t;
    }
}

template <int MODE, int GLOG2>
static hipError_t launch_rg_t(const BatchTables &t, const float *x, const float *selfq, float *out, float eps, hipStream_t s)
{
    const Options &o = options();
    constexpr int RPI = 64 >> GLOG2;
    constexpr bool pna = MODE == GNNB_AGG_PNA;
    const int cus = device_cu_count();
    // row-instructions in flight per wave and batch, and the workgroups (of four waves) per CU that the register
    // allocation of that depth leaves room for (R = 2: 94 VGPRs = five waves per SIMD; PNA keeps four accumulators per row)
    // (measured at BASELINE configs 2 / 4 / 5: many short-lived waves of ONE row-instruction per batch beat fewer, deeper ones
    // at equal rows in flight -- GCN 14.7 us at R = 1 x 16 workgroups per CU against 16.1 / 15.8 / 17.3 at R = 2 / 3 / 4 with
    // what is resident; PNA 67.4 at 32 workgroups per CU against 71.8 at 16 and 76.5 at 8)
    int R = o.agg_rg_r > 0 ? (int)o.agg_rg_r : 1;
    R = std::min(std::max(R, 1), pna ? 2 : 4);
    static const int wgs_of_r[5] = {0, 32, 16, 3, 2};
    int wgs = o.agg_rg_wgs > 0 ? (int)o.agg_rg_wgs : wgs_of_r[R];
    wgs = std::min(std::max(wgs, 1), 64); // (beyond what is resident: short-lived workgroups handed out by the dispatcher)
    int grid = cus * wgs;
    // every wave owns a contiguous run of rpw rows: whole batches of R row-instructions
    int rpw = (t.num_nodes + grid * 4 - 1) / (grid * 4);
    rpw = ((rpw + R * RPI - 1) / (R * RPI)) * (R * RPI);
    grid = std::min(grid, (t.num_nodes + 4 * rpw - 1) / (4 * rpw));
    if (grid >= 8)
        grid = (grid + 7) & ~7; // (whole XCD rounds: the kernel's row dealing wants gridDim % 8 == 0; empty waves return at once)
    const int flags = o.agg_rg_flags & 15;
#define GNNB_RG_LAUNCH(RV, FV, MB)                                                                                         \
    hipLaunchKernelGGL((k_aggregate_rg<MODE, GLOG2, RV, FV, MB>), dim3(grid), dim3(256), 0, s, x, selfq, out, t.node_rec, t.col, \
                       t.dinv, t.gcoef, t.num_nodes, rpw, eps)
#define GNNB_RG_R(RV, MB)                                                        \
    do {                                                                         \
        bool done = true;                                                        \
        if constexpr (MODE == GNNB_AGG_GCN && GLOG2 == 5 && RV == 1) {           \
            switch (flags) { /* the ablations exist for the north-star shape */  \
            case 4: GNNB_RG_LAUNCH(RV, 4, MB); break;                            \
            case 5: GNNB_RG_LAUNCH(RV, 5, MB); break;                            \
            case 8: GNNB_RG_LAUNCH(RV, 8, MB); break;                            \
            case 9: GNNB_RG_LAUNCH(RV, 9, MB); break;                            \
            case 12: GNNB_RG_LAUNCH(RV, 12, MB); break;                          \
            default: done = false; break;                                        \
            }                                                                    \
        } else {                                                                 \
            done = false;                                                        \
        }                                                                        \
        if (!done) {                                                             \
            switch (flags & 3) {                                                 \
            case 0: GNNB_RG_LAUNCH(RV, 0, MB); break;                            \
            case 1: GNNB_RG_LAUNCH(RV, 1, MB); break;                            \
            case 2: GNNB_RG_LAUNCH(RV, 2, MB); break;                            \
            default: GNNB_RG_LAUNCH(RV, 3, MB); break;                           \
            }                                                                    \
        }                                                                        \
    } while (0)
    if constexpr (pna) {
        if (R <= 1)
            GNNB_RG_R(1, 8);
        else
            GNNB_RG_R(2, 4);
    } else {
        switch (R) {
        case 1: GNNB_RG_R(1, 8); break;
        case 2: GNNB_RG_R(2, 5); break;
        case 3: GNNB_RG_R(3, 3); break;
        default: GNNB_RG_R(4, 2); break;
        }
    }
#undef GNNB_RG_R
#undef GNNB_RG_LAUNCH
    return hipGetLastError();
}

// hipErrorNotSupported (nothing launched): the caller runs the ring form
hipError_t launch_aggregate_rg(const BatchTables &t, int kind, const float *x, const float *selfq, float *out, int width,
                               float eps, hipStream_t s)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    if (t.tile_lo != 0) // (a large segment's row range starts at a tile the host does not know the row of: ring form)
        return hipErrorNotSupported;
    if (width != 64 && width != 128 && width != 256)
        return hipErrorNotSupported;
    if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)selfq) & 15) != 0)
        return hipErrorNotSupported;
#define GNNB_RG_W(K)                                                                 \
    case K:                                                                          \
        return width == 64    ? launch_rg_t<K, 4>(t, x, selfq, out, eps, s)          \
               : width == 128 ? launch_rg_t<K, 5>(t, x, selfq, out, eps, s)          \
                              : launch_rg_t<K, 6>(t, x, selfq, out, eps, s);
    switch (kind) {
        GNNB_RG_W(GNNB_AGG_GCN)
        GNNB_RG_W(GNNB_AGG_SUM)
        GNNB_RG_W(GNNB_AGG_MEAN)
        GNNB_RG_W(GNNB_AGG_SIMPLE)
        GNNB_RG_W(GNNB_AGG_PNA)
    default:
        return hipErrorNotSupported;
    }
#undef GNNB_RG_W
}

} // namespace gnnb
