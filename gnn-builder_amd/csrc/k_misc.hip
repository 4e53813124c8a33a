// k_misc.hip -- output activation, fixed-point grid, probe read-back
// Part of libgnnb_hip.so (hand-written gfx950 / CDNA4 kernels of the GNNBuilder hot path); wavefront = 64 lanes.
#include "gnnb_device.h"

namespace gnnb {

// GNNModel.output_activation (models.py:500-502, 572-573): softmax / log_softmax over each graph's output row.
// OUT is a handful of values (1..19 at the BASELINE configs): one lane per graph, three passes over the row.
__global__ __launch_bounds__(WG) void k_output_activation(float *__restrict__ out, int B, int n, int kind)
{
    const int g = blockIdx.x * WG + threadIdx.x;
    if (g >= B)
        return;
    float *o = out + (size_t)g * n;
    float mx = o[0];
    for (int i = 1; i < n; i++)
        mx = fmaxf(mx, o[i]);
    float sum = 0.0f;
    for (int i = 0; i < n; i++)
        sum += expf(o[i] - mx);
    for (int i = 0; i < n; i++)
        o[i] = kind == GNNB_OUT_SOFTMAX ? expf(o[i] - mx) / sum : (o[i] - mx) - logf(sum);
}

hipError_t launch_output_activation(float *out, int num_graphs, int n, int kind, hipStream_t s)
{
    if (num_graphs <= 0 || kind == GNNB_OUT_NONE)
        return hipSuccess;
    hipLaunchKernelGGL(k_output_activation, dim3((num_graphs + WG - 1) / WG), dim3(WG), 0, s, out, num_graphs, n, kind);
    return hipGetLastError();
}

// ---- degree classes (PNA under a degree promise <= GNNB_DEG_CLASSES): the batch's rows sorted by in-degree (0 .. 15) into
// 128-row tiles of ONE class each, for k_linear_dma's row-class mode (k_gemm.hip).  perm[position] = row (-1: padding),
// tile_cls[tile] = class = in-degree.  A STABLE counting sort without atomics (same-address atomics from every wave of the batch
// cost 150 us per pass at BASELINE config 4): GNNB_DEG_RUNS waves take one contiguous run of rows each -- count per (run,
// class) -> one workgroup turns the counts into bases (classes padded to whole tiles, runs in order) -> every wave places its
// rows in order.  Rows keep their order inside a class, so a tile's rows are near each other in memory.
static constexpr int GNNB_DEG_RUNS = 256;
__device__ __forceinline__ int deg_class_of(const int32_t *__restrict__ row_ptr, int v) { return min(max(row_ptr[v + 1] - row_ptr[v], 0), GNNB_DEG_MAX); }

__global__ __launch_bounds__(WG) void k_deg_count(const int32_t *__restrict__ row_ptr, int N, int promise, int32_t *__restrict__ work,
                                                  int32_t *__restrict__ err, int32_t *__restrict__ err_host)
{
    const int run = blockIdx.x * (WG / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per = ((N + GNNB_DEG_RUNS - 1) / GNNB_DEG_RUNS + 63) & ~63; // rows per run (whole 64-row steps)
    const int r0 = min(run * per, N), r1 = min(r0 + per, N);
    int cnt = 0; // lane c < GNNB_DEG_CLASSES: this run's rows of class c
    bool bad = false;
    for (int v0 = r0; v0 < r1; v0 += 64) {
        const int v = v0 + lane;
        int c = -1;
        if (v < r1) {
            const int d = row_ptr[v + 1] - row_ptr[v];
            bad |= d > promise || d > GNNB_DEG_MAX;
            c = min(max(d, 0), GNNB_DEG_MAX);
        }
#pragma unroll
        for (int k = 0; k < GNNB_DEG_CLASSES; k++) {
            const int n = __builtin_popcountll(__ballot(c == k));
            if (lane == k)
                cnt += n;
        }
    }
    if (lane < 16)
        work[run * 16 + lane] = lane < GNNB_DEG_CLASSES ? cnt : 0;
    if (__ballot(bad) && lane == 0) { // the caller's max_degree promise is broken: flagged, results unspecified
        atomicOr(err, 32);
        if (err_host)
            *reinterpret_cast<volatile int32_t *>(err_host) = 32;
    }
}

__global__ __launch_bounds__(GNNB_DEG_RUNS) void k_deg_offsets(int32_t *__restrict__ work, int32_t *__restrict__ tile_cls, int max_tiles)
{
    // one workgroup, thread r = run r: per class an exclusive scan of the runs' counts (wave scan + the waves' totals through
    // LDS), the classes' starts padded to whole tiles; every run's count is replaced by its base
    static_assert(GNNB_DEG_RUNS % 64 == 0 && GNNB_DEG_RUNS <= 1024, "one thread per run");
    __shared__ int wtot[GNNB_DEG_RUNS / 64];
    __shared__ int cls_start[GNNB_DEG_CLASSES + 1]; // in tiles
    const int r = threadIdx.x, lane = r & 63, wave = r >> 6;
    int cnt[GNNB_DEG_CLASSES];
#pragma unroll
    for (int c = 0; c < GNNB_DEG_CLASSES; c++)
        cnt[c] = work[r * 16 + c];
    int tiles_before = 0;
#pragma unroll 1
    for (int c = 0; c < GNNB_DEG_CLASSES; c++) {
        int n = 0;
#pragma unroll
        for (int k = 0; k < GNNB_DEG_CLASSES; k++) // (static indexing of the register array)
            n = k == c ? cnt[k] : n;
        int incl = n; // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d)
                incl += o;
        }
        if (lane == 63)
            wtot[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < GNNB_DEG_RUNS / 64; w++) {
            const int t = wtot[w];
            before += w < wave ? t : 0;
            total += t;
        }
        __syncthreads();
        work[r * 16 + c] = tiles_before * 128 + before + incl - n;
        if (r == 0)
            cls_start[c] = tiles_before;
        tiles_before += (total + 127) / 128;
    }
    if (r == 0)
        cls_start[GNNB_DEG_CLASSES] = tiles_before;
    __syncthreads();
    for (int t = r; t < max_tiles; t += GNNB_DEG_RUNS) {
        int c = 0; // (tiles behind the last class: all padding, any class)
        for (int k = 0; k < GNNB_DEG_CLASSES; k++)
            if (t >= cls_start[k] && t < cls_start[k + 1])
                c = k;
        tile_cls[t] = c;
    }
}

__global__ __launch_bounds__(WG) void k_deg_scatter(const int32_t *__restrict__ row_ptr, int N, const int32_t *__restrict__ work,
                                                    int32_t *__restrict__ perm, int max_pos)
{
    const int run = blockIdx.x * (WG / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per = ((N + GNNB_DEG_RUNS - 1) / GNNB_DEG_RUNS + 63) & ~63;
    const int r0 = min(run * per, N), r1 = min(r0 + per, N);
    int cur = lane < GNNB_DEG_CLASSES ? work[run * 16 + lane] : 0; // lane c: the next position of class c
    for (int v0 = r0; v0 < r1; v0 += 64) {
        const int v = v0 + lane;
        const int c = v < r1 ? deg_class_of(row_ptr, v) : -1;
        int pos = -1;
#pragma unroll
        for (int k = 0; k < GNNB_DEG_CLASSES; k++) {
            const unsigned long long m = __ballot(c == k);
            const int base = __builtin_amdgcn_readlane(cur, k);
            if (c == k)
                pos = base + __builtin_popcountll(m & ((1ull << lane) - 1));
            if (lane == k)
                cur += __builtin_popcountll(m);
        }
        if (pos >= 0 && pos < max_pos)
            perm[pos] = v;
    }
}

hipError_t launch_degree_classes(const BatchTables &t, int promise, int32_t *work, int32_t *perm, int32_t *tile_cls, int max_tiles,
                                 hipStream_t s)
{
    if (t.num_nodes <= 0)
        return hipSuccess;
    hipError_t e = hipMemsetAsync(perm, 0xFF, (size_t)max_tiles * 128 * sizeof(int32_t), s);
    if (e != hipSuccess)
        return e;
    const int grid = GNNB_DEG_RUNS / (WG / 64);
    hipLaunchKernelGGL(k_deg_count, dim3(grid), dim3(WG), 0, s, t.row_ptr, t.num_nodes, promise, work, t.err, t.err_host_dev);
    hipLaunchKernelGGL(k_deg_offsets, dim3(1), dim3(GNNB_DEG_RUNS), 0, s, work, tile_cls, max_tiles);
    hipLaunchKernelGGL(k_deg_scatter, dim3(grid), dim3(WG), 0, s, t.row_ptr, t.num_nodes, work, perm, max_tiles * 128);
    return hipGetLastError();
}

// ap_fixed<W, I, AP_TRN, AP_WRAP> grid (reference code_gen.py:39-52, model.h.jinja:41-45): truncate towards minus
// infinity to a multiple of 2^-(W-I), wrap into [-2^(I-1), 2^(I-1)).  `inv_step` = 2^(W-I), `span` = 2^I.
__global__ __launch_bounds__(WG) void k_quantize(const float *__restrict__ src, float *__restrict__ dst, size_t n,
                                                 float inv_step, float step, float half_span, float span)
{
    for (size_t i = blockIdx.x * (size_t)WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
        float v = floorf(src[i] * inv_step) * step;
        v = v - span * floorf((v + half_span) / span); // two's-complement wrap
        dst[i] = v;
    }
}

hipError_t launch_quantize(const float *src, float *dst, size_t n, int W, int I, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const float inv_step = ldexpf(1.0f, W - I), step = ldexpf(1.0f, -(W - I));
    const float span = ldexpf(1.0f, I), half = ldexpf(1.0f, I - 1);
    const unsigned grid = (unsigned)std::min<size_t>((n + WG - 1) / WG, 4096);
    hipLaunchKernelGGL(k_quantize, dim3(grid), dim3(WG), 0, s, src, dst, n, inv_step, step, half, span);
    return hipGetLastError();
}

#ifdef GNNB_PROBE
extern "C" int gnnb_probe_read(unsigned long long *host, int count)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe), sizeof(unsigned long long) * count);
}
#endif


} // namespace gnnb
