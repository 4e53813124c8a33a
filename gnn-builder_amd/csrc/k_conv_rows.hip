// k_conv_rows.hip -- one GCN / GIN conv layer for a RANGE of node rows, small-footprint form (large segment of a batch)
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include "gnnb_device.h"

namespace gnnb {

// =====================================================================================
// k_conv_rows: aggregate + update of one conv layer for node rows [row_lo, N), 16 rows per workgroup
// =====================================================================================
// Reference: gcn_conv (gnn_builder_lib.h:1213-1387: normalised aggregate, `linear`, activation) and gin_conv
// (:1389-1544: (1 + eps) x_i + sum_j x_j, `linear`, ReLU, `linear`, activation), with the skip connection the generated
// compute_gnn_head adds on the middle layers (templates/model.cpp.jinja:264-311).
//
// Why a second form of the layer-by-layer path.  The graphs of a batch's LARGE SEGMENT (gnnb_workspace_set_large_segment:
// the few molecules beyond the stage capacity of the LDS-resident stack kernels, ~1 graph in 100 of ogbg-molhiv, ~4 % of
// the rows) cannot be staged whole, so they go layer by layer.  Through the big kernels (ring aggregate: 160 KB of LDS per
// CU; weights-in-LDS GEMM: 64 KB) that is eleven launches of ~9 us each on 4 k rows, run BEHIND the stack kernel of the
// rest of the batch: +100 us on a 250 us step.  This kernel is built to run BESIDE the stack kernel instead, on a forked
// stream: 256 threads, one 16 x 132 fp32 tile of LDS (8.4 KB: the stack kernels leave 9.4 KB or more of a CU free),
// under 96 registers (one wave slot per SIMD is left), neighbour rows and weights fetched straight from L2 (the previous
// layer's rows were just written, the weights are shared by every workgroup).  One launch per layer:
//   1. aggregate: 16 lanes per row gather the row's in-neighbours (node record: first four sources inline, the rest of
//      the CSR row from `col`) in CSR order, self term last -> LDS tile A [16][K]
//   2. update: Y = A . W^T + b on v_mfma_f32_16x16x4_f32, wave w takes the 16-column slices w, w + 4, ...; A fragments
//      from LDS, W fragments from L2
//   3. GCN: + skip, activation, store.  GIN: ReLU -> the LDS tile (in place, behind a barrier), second product with W2,
//      + b2, + skip, activation, store.
// Widths: K, N <= 128 (the tile), any K (zero padded to whole MFMA k blocks); float4 fetches when K % 4 == 0.
static constexpr int CR_THREADS = 256;
static constexpr int CR_ROWS = 16;
static constexpr int CR_MAXW = 128;
static constexpr int CR_LD = CR_MAXW + 4; // padded LDS row (floats): conflict-free fragment reads

struct ConvRowsArgs {
    const float *x;        // [N, K] layer input (batch-global rows)
    float *y;              // [N, Nout] layer output
    const float *skip;     // [N, Nout] or nullptr
    const int4 *node_rec;  // [2N]
    const int32_t *col;    // [E]
    const float *dinv;     // [N]
    const float *w1, *b1;  // [N1, K]: GCN N1 = Nout; GIN N1 = Nout (hidden = out, reference models.py:90)
    const float *w2, *b2;  // GIN: [Nout, Nout]
    int row_lo, N, K, Nout;
    int gin;               // 0 = GCN, 1 = GIN
    float eps;
};

// Y[16][n] = T[16][k] . W[n][k]^T for the wave's column slices; fn(column nn, accumulator rows r -> tile row 4 lg + r)
template <typename F>
__device__ __forceinline__ void cr_product(const float *__restrict__ T, const float *__restrict__ W, int k, int n, int wave,
                                           int li, int lg, bool vecw, F &&fn)
{
    const int kpad = (k + 15) & ~15;
    for (int sl = wave; sl * 16 < n; sl += CR_THREADS / 64) {
        const int nn = sl * 16 + li;
        const int nnc = nn < n ? nn : n - 1;
        const float *wrow = W + (size_t)nnc * k;
        const float *arow = T + li * CR_LD;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < kpad; kb += 32) {
            float4 a[2], w[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int kk = kb + 16 * u + 4 * lg;
                a[u] = kk < kpad ? *reinterpret_cast<const float4 *>(arow + kk) : make_float4(0.f, 0.f, 0.f, 0.f); // (tile zero padded to kpad)
                if (vecw)
                    w[u] = kk < k ? *reinterpret_cast<const float4 *>(wrow + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
                else
                    w[u] = make_float4(kk < k ? wrow[kk] : 0.f, kk + 1 < k ? wrow[kk + 1] : 0.f, kk + 2 < k ? wrow[kk + 2] : 0.f,
                                       kk + 3 < k ? wrow[kk + 3] : 0.f);
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].x, w[0].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1].x, w[1].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].y, w[0].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1].y, w[1].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].z, w[0].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1].z, w[1].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0].w, w[0].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1].w, w[1].w, acc1, 0, 0, 0);
        }
        if (nn < n) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++)
                v[r] = acc0[r] + acc1[r];
            fn(nn, v);
        }
    }
}

template <int ACT>
__global__ __launch_bounds__(CR_THREADS, 5) void k_conv_rows(ConvRowsArgs p)
{
    __shared__ __attribute__((aligned(16))) float T[CR_ROWS * CR_LD];
    __builtin_amdgcn_s_setprio(GNNB_GUEST_PRIO); // (co-runs with the conv-stack kernel of the rest of the batch: see k_graph_prep)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int r0 = p.row_lo + blockIdx.x * CR_ROWS;
    const int K = p.K, kpad = (K + 15) & ~15;

    // ---- 1. aggregate the 16 rows into the tile: 16 lanes per row, CSR order, self term last
    {
        const int r = tid >> 4, l16 = tid & 15;
        const int v = r0 + r;
        const bool ok = v < p.N;
        const int vc = ok ? v : p.N - 1;
        const int4 rec0 = p.node_rec[2 * (size_t)vc], rec1 = p.node_rec[2 * (size_t)vc + 1];
        const int rp0 = rec0.x, deg = ok ? rec0.y : 0;
        const int jn[4] = {rec0.z, rec0.w, rec1.x, rec1.y};
        const float di = p.gin ? 1.0f : p.dinv[vc];
        const float cself = p.gin ? 1.0f + p.eps : di * di;
        float cj[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            cj[q] = deg > q ? (p.gin ? 1.0f : di * p.dinv[jn[q]]) : 0.0f; // (unused slots alias the row itself: in range)
        if ((K & 3) == 0) {
            for (int k4 = l16; k4 * 4 < kpad; k4 += 16) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k4 * 4 < K && ok) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 xv = *reinterpret_cast<const float4 *>(p.x + (size_t)jn[q] * K + k4 * 4);
                        acc.x += cj[q] * xv.x, acc.y += cj[q] * xv.y, acc.z += cj[q] * xv.z, acc.w += cj[q] * xv.w;
                    }
                    for (int e = rp0 + 4; e < rp0 + deg; e++) {
                        const int j = p.col[e];
                        const float c = p.gin ? 1.0f : di * p.dinv[j];
                        const float4 xv = *reinterpret_cast<const float4 *>(p.x + (size_t)j * K + k4 * 4);
                        acc.x += c * xv.x, acc.y += c * xv.y, acc.z += c * xv.z, acc.w += c * xv.w;
                    }
                    const float4 xs = *reinterpret_cast<const float4 *>(p.x + (size_t)v * K + k4 * 4);
                    acc.x += cself * xs.x, acc.y += cself * xs.y, acc.z += cself * xs.z, acc.w += cself * xs.w;
                }
                *reinterpret_cast<float4 *>(T + r * CR_LD + k4 * 4) = acc;
            }
        } else {
            for (int k = l16; k < kpad; k += 16) {
                float acc = 0.0f;
                if (k < K && ok) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        acc += cj[q] * p.x[(size_t)jn[q] * K + k];
                    for (int e = rp0 + 4; e < rp0 + deg; e++) {
                        const int j = p.col[e];
                        acc += (p.gin ? 1.0f : di * p.dinv[j]) * p.x[(size_t)j * K + k];
                    }
                    acc += cself * p.x[(size_t)v * K + k];
                }
                T[r * CR_LD + k] = acc;
            }
        }
    }
    __syncthreads();

    const int n = p.Nout;
    const bool vec1 = (K & 3) == 0 && (((uintptr_t)p.w1) & 15) == 0;
    auto finish = [&](int nn, const float (&v)[4], const float *bias) { // + bias, + skip, activation, store
        const float bv = bias ? bias[nn] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = r0 + lg * 4 + r;
            if (row < p.N) {
                float o = v[r] + bv;
                if (p.skip)
                    o += p.skip[(size_t)row * n + nn];
                p.y[(size_t)row * n + nn] = act_t<ACT>(o);
            }
        }
    };
    if (!p.gin) {
        // ---- 2./3. GCN: Y = act(A . W^T + b (+ skip))
        cr_product(T, p.w1, K, n, wave, li, lg, vec1, [&](int nn, const float (&v)[4]) { finish(nn, v, p.b1); });
        return;
    }
    // ---- GIN: hidden = relu(A . W1^T + b1) -> the tile (in place: everybody has read A first), then the second linear
    float hid[2][4];
    int hcol[2] = {-1, -1}, nh = 0;
    cr_product(T, p.w1, K, n, wave, li, lg, vec1, [&](int nn, const float (&v)[4]) {
        const float bv = p.b1 ? p.b1[nn] : 0.0f;
        if (nh < 2) {
            hcol[nh] = nn;
#pragma unroll
            for (int r = 0; r < 4; r++)
                hid[nh][r] = fmaxf(v[r] + bv, 0.0f);
            nh++;
        }
    });
    __syncthreads();
    // (a wave owns at most two slices of a tile <= 128 wide; columns n .. npad - 1 of the tile are zeroed for the k padding)
#pragma unroll
    for (int s = 0; s < 2; s++)
        if (hcol[s] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                T[(lg * 4 + r) * CR_LD + hcol[s]] = hid[s][r];
        }
    {
        const int npad = (n + 15) & ~15;
        for (int i = tid; i < CR_ROWS * (npad - n); i += CR_THREADS)
            T[(i / (npad - n)) * CR_LD + n + i % (npad - n)] = 0.0f;
    }
    __syncthreads();
    const bool vec2 = (n & 3) == 0 && (((uintptr_t)p.w2) & 15) == 0;
    cr_product(T, p.w2, n, n, wave, li, lg, vec2, [&](int nn, const float (&v)[4]) { finish(nn, v, p.b2); });
}

// hipErrorNotSupported when the layer does not suit this form (caller takes the big layer-by-layer kernels)
hipError_t launch_conv_rows(const BatchTables &t, int conv_type, const float *x, int K, const float *w1, const float *b1,
                            const float *w2, const float *b2, int Nout, const float *skip, float *y, int row_lo, int act,
                            float eps, hipStream_t s)
{
    if (conv_type != GNNB_CONV_GCN && conv_type != GNNB_CONV_GIN)
        return hipErrorNotSupported;
    if (K < 1 || K > CR_MAXW || Nout < 1 || Nout > CR_MAXW)
        return hipErrorNotSupported;
    if ((K & 3) == 0 && (((uintptr_t)x) & 15))
        return hipErrorNotSupported;
    const int M = t.num_nodes - row_lo;
    if (M <= 0)
        return hipSuccess;
    ConvRowsArgs p;
    p.x = x;
    p.y = y;
    p.skip = skip;
    p.node_rec = t.node_rec;
    p.col = t.col;
    p.dinv = t.dinv;
    p.w1 = w1;
    p.b1 = b1;
    p.w2 = w2;
    p.b2 = b2;
    p.row_lo = row_lo;
    p.N = t.num_nodes;
    p.K = K;
    p.Nout = Nout;
    p.gin = conv_type == GNNB_CONV_GIN ? 1 : 0;
    p.eps = eps;
    const unsigned grid = (unsigned)((M + CR_ROWS - 1) / CR_ROWS);
    auto go = [&](auto atag) {
        constexpr int ACT = decltype(atag)::value;
        hipLaunchKernelGGL(k_conv_rows<ACT>, dim3(grid), dim3(CR_THREADS), 0, s, p);
    };
    GNNB_DISPATCH_ACT(act, go)
    return hipGetLastError();
}

} // namespace gnnb
