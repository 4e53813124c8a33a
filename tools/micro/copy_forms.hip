// Calibration of THIS pool's HBM ceiling: what do plain byte-moving kernels reach on the box that runs bench.py?
// (the gather-aggregate kernel sits at 0.61 of 8 TB/s and the argument that it cannot go further
//  rests on a float4 copy that itself measures 10-17 % below /opt/skills/guides/MI355X_MICROARCH.md's 6.29 TB/s.)
// Forms (all 16 B per lane, grid-stride over float4 elements, `wgs` workgroups of 256 threads):
//   copy        plain global_load_dwordx4 -> global_store_dwordx4
//   copy_nt     non-temporal loads and stores (__builtin_nontemporal_*: `nt` bit)
//   copy_nt_st  plain loads, non-temporal stores (the aggregate kernels' choice)
//   copy_u4     as copy, four independent 16-B loads in flight per lane before the first store
//   read        read-only sweep (sum folded into one conditional store that never fires)
//   write       write-only sweep
//   write_nt    write-only, non-temporal
//   copy_lds    global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction) -> ds_read_b128 -> global_store_dwordx4,
//               a ring of four 1-KiB slots per wave, no workgroup barrier
// Sizes: bytes MOVED per launch (read + written) = 76 MB, 431 MB, 1.7 GB (the C2 / C5 aggregate launches and 4x that);
// buffers rotate over > 256 MiB so that nothing is served by the Infinity Cache (the "HBM regime" of bench.py).
// Timing: HIP events around `reps` back-to-back launches on one stream, best and median of `trials`.
// build: hipcc -O3 --offload-arch=gfx950 -o bin/copy_forms copy_forms.hip      (tools/copy_forms.py does this)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));          \
            exit(2);                                                                                  \
        }                                                                                             \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_copy(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy_nt(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ __launch_bounds__(256) void k_copy_nt_st(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        __builtin_nontemporal_store(src[i], dst + i);
}
__global__ __launch_bounds__(256) void k_copy_u4(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride);
        __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n; i += stride)
        __builtin_nontemporal_store(src[i], dst + i);
}
__global__ __launch_bounds__(256) void k_read(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        s += a + b + c + d;
    }
    for (; i < n; i += stride)
        s += src[i];
    if (s[0] + s[1] + s[2] + s[3] == 123456.789f) // (never: the inputs are zeros)
        dst[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_write(const f32x4 *__restrict__, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        dst[i] = v;
}
__global__ __launch_bounds__(256) void k_write_nt(const f32x4 *__restrict__, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        __builtin_nontemporal_store(v, dst + i);
}
// LDS-DMA copy: a wave moves 1-KiB pieces (64 lanes x 16 B) through its OWN ring of four 1-KiB LDS slots (no workgroup
// barrier), three pieces in flight beside the one being stored.  M0 = LDS byte address of the slot (wave-uniform); the
// instruction adds lane * 16 itself.  vmcnt counts DMA pieces and stores alike, in issue order: at the wait of piece i the
// operations younger than DMA(i) are DMA(i+1), st(i-2), DMA(i+2), st(i-1), DMA(i+3) -- vmcnt(3) is the safe, slightly strict bound.
__global__ __launch_bounds__(256) void k_copy_lds(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    __shared__ __attribute__((aligned(16))) char buf[4 * 4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nwaves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + wave;
    const size_t pieces = n / 64; // (n is a multiple of 64 float4: the host rounds)
    char *mine = buf + wave * 4096;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)mine;
    auto fire = [&](size_t p, int slot) {
        const f32x4 *g = src + p * 64 + lane;
        const uint32_t m0 = __builtin_amdgcn_readfirstlane(lbase + slot * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(g) : "memory");
    };
    size_t p = w;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (p + k * nwaves < pieces)
            fire(p + k * nwaves, k);
    int slot = 0;
    for (; p < pieces; p += nwaves) {
        const size_t pn = p + 3 * nwaves;
        if (pn < pieces) {
            fire(pn, (slot + 3) & 3);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4 v = *reinterpret_cast<const f32x4 *>(mine + slot * 1024 + lane * 16);
        __builtin_nontemporal_store(v, dst + p * 64 + lane);
        slot = (slot + 1) & 3;
    }
}

// ---- the same bytes with BLOCKED ownership (what a persistent kernel that walks a contiguous run of rows per workgroup does --
// k_aggregate_ring: one run of node tiles per CU): workgroup b owns elements [b n / G, (b + 1) n / G).  At any moment the chip then
// reads G windows that lie n / G apart instead of one sliding window of G x 4 KiB: does the address pattern alone cost bandwidth?
__global__ __launch_bounds__(256) void k_copy_nt_blk(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
// LDS-DMA ring per wave (as k_copy_lds), the workgroup owning a contiguous run of 1-KiB pieces, its four waves interleaved inside it
__global__ __launch_bounds__(256) void k_copy_lds_blk(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    __shared__ __attribute__((aligned(16))) char buf[4 * 4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pieces = n / 64;
    const size_t per = (pieces + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < pieces ? lo + per : pieces;
    char *mine = buf + wave * 4096;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)mine;
    auto fire = [&](size_t p, int slot) {
        const f32x4 *g = src + p * 64 + lane;
        const uint32_t m0 = __builtin_amdgcn_readfirstlane(lbase + slot * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(g) : "memory");
    };
    size_t p = lo + wave;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (p + k * 4 < hi)
            fire(p + k * 4, k);
    int slot = 0;
    for (; p < hi; p += 4) {
        const size_t pn = p + 12;
        if (pn < hi) {
            fire(pn, (slot + 3) & 3);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4 v = *reinterpret_cast<const f32x4 *>(mine + slot * 1024 + lane * 16);
        __builtin_nontemporal_store(v, dst + p * 64 + lane);
        slot = (slot + 1) & 3;
    }
}
// ... and with 1024-thread workgroups (16 waves, as the ring kernel): the workgroup owns a contiguous run, 16 waves interleaved
__global__ __launch_bounds__(1024) void k_copy_lds_blk16(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    __shared__ __attribute__((aligned(16))) char buf[16 * 4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pieces = n / 64;
    const size_t per = (pieces + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < pieces ? lo + per : pieces;
    char *mine = buf + wave * 4096;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)mine;
    auto fire = [&](size_t p, int slot) {
        const f32x4 *g = src + p * 64 + lane;
        const uint32_t m0 = __builtin_amdgcn_readfirstlane(lbase + slot * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(g) : "memory");
    };
    size_t p = lo + wave;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (p + k * 16 < hi)
            fire(p + k * 16, k);
    int slot = 0;
    for (; p < hi; p += 16) {
        const size_t pn = p + 48;
        if (pn < hi) {
            fire(pn, (slot + 3) & 3);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4 v = *reinterpret_cast<const f32x4 *>(mine + slot * 1024 + lane * 16);
        __builtin_nontemporal_store(v, dst + p * 64 + lane);
        slot = (slot + 1) & 3;
    }
}
// the ring kernel's OTHER property: 16-wave workgroups whose waves interleave over the whole buffer (no blocked ownership)
__global__ __launch_bounds__(1024) void k_copy_lds_16(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    __shared__ __attribute__((aligned(16))) char buf[16 * 4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nwaves = (size_t)gridDim.x * 16, w = (size_t)blockIdx.x * 16 + wave;
    const size_t pieces = n / 64;
    char *mine = buf + wave * 4096;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)mine;
    auto fire = [&](size_t p, int slot) {
        const f32x4 *g = src + p * 64 + lane;
        const uint32_t m0 = __builtin_amdgcn_readfirstlane(lbase + slot * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(g) : "memory");
    };
    size_t p = w;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (p + k * nwaves < pieces)
            fire(p + k * nwaves, k);
    int slot = 0;
    for (; p < pieces; p += nwaves) {
        const size_t pn = p + 3 * nwaves;
        if (pn < pieces) {
            fire(pn, (slot + 3) & 3);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4 v = *reinterpret_cast<const f32x4 *>(mine + slot * 1024 + lane * 16);
        __builtin_nontemporal_store(v, dst + p * 64 + lane);
        slot = (slot + 1) & 3;
    }
}

// ... and in between: the 16-wave workgroup takes CHUNKS of CK KiB round robin (chunk c = b, b + G, ...), its waves interleaved inside
// a chunk -- ownership interleaved at the granularity of one LDS stage of the ring kernel instead of one run per launch
template <int CK>
__global__ __launch_bounds__(1024) void k_copy_lds_chunk16(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    __shared__ __attribute__((aligned(16))) char buf[16 * 4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pieces = n / 64;
    char *mine = buf + wave * 4096;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)mine;
    auto fire = [&](size_t p, int slot) {
        const f32x4 *g = src + p * 64 + lane;
        const uint32_t m0 = __builtin_amdgcn_readfirstlane(lbase + slot * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(g) : "memory");
    };
    // the wave's k-th piece: chunk (b + (k / PW) G), piece wave + 16 (k % PW) inside it; PW = pieces per wave and chunk
    constexpr int PW = CK / 16;
    const size_t nchunks = (pieces + CK - 1) / CK;
    auto piece_of = [&](size_t k) -> size_t {
        const size_t c = blockIdx.x + (k / PW) * (size_t)gridDim.x;
        return c < nchunks ? c * CK + wave + 16 * (k % PW) : pieces;
    };
    size_t k = 0;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (piece_of(j) < pieces)
            fire(piece_of(j), j);
    int slot = 0;
    for (;; k++) {
        const size_t p = piece_of(k);
        if (p >= pieces)
            break;
        const size_t pn = piece_of(k + 3);
        if (pn < pieces) {
            fire(pn, (slot + 3) & 3);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4 v = *reinterpret_cast<const f32x4 *>(mine + slot * 1024 + lane * 16);
        __builtin_nontemporal_store(v, dst + p * 64 + lane);
        slot = (slot + 1) & 3;
    }
}

typedef void (*kern_t)(const f32x4 *, f32x4 *, size_t);
struct Form {
    const char *name;
    kern_t k;
    int reads, writes; // bytes moved per element = 16 * (reads + writes)
    int threads = 256; // per workgroup
};

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20, trials = argc > 2 ? atoi(argv[2]) : 5;
    const Form forms[] = {{"copy", k_copy, 1, 1},         {"copy_nt", k_copy_nt, 1, 1}, {"copy_nt_st", k_copy_nt_st, 1, 1},
                          {"copy_u4", k_copy_u4, 1, 1},   {"read", k_read, 1, 0},       {"write", k_write, 0, 1},
                          {"write_nt", k_write_nt, 0, 1}, {"copy_lds", k_copy_lds, 1, 1},
                          {"copy_nt_blk", k_copy_nt_blk, 1, 1}, {"copy_lds_blk", k_copy_lds_blk, 1, 1},
                          {"copy_lds_16", k_copy_lds_16, 1, 1, 1024}, {"copy_lds_blk16", k_copy_lds_blk16, 1, 1, 1024},
                          {"copy_lds_chunk16_64k", k_copy_lds_chunk16<64>, 1, 1, 1024}, {"copy_lds_chunk16_128k", k_copy_lds_chunk16<128>, 1, 1, 1024},
                          {"copy_lds_chunk16_256k", k_copy_lds_chunk16<256>, 1, 1, 1024}};
    const double moved[] = {76.2e6, 431.3e6, 1725.2e6};
    const int wgs[] = {256, 512, 1024, 2048, 4096, 16384};
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    // one arena of 2.5 GiB for sources and one for destinations; launch r of a loop uses slice (r mod slices)
    const size_t arena = (size_t)2560 << 20;
    char *A, *B;
    CK(hipMalloc(&A, arena));
    CK(hipMalloc(&B, arena));
    CK(hipMemset(A, 0, arena));
    CK(hipMemset(B, 0, arena));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // every copy form moves the bytes it claims: a patterned 4-MiB source, compared on the host
    {
        const size_t vn = (size_t)4 << 16; // float4 elements
        std::vector<uint32_t> pat(vn * 4), back(vn * 4);
        for (size_t i = 0; i < pat.size(); i++)
            pat[i] = (uint32_t)(i * 2654435761u);
        for (const Form &f : forms) {
            if (!(f.reads && f.writes))
                continue;
            CK(hipMemcpy(A, pat.data(), vn * 16, hipMemcpyHostToDevice));
            CK(hipMemset(B, 0xff, vn * 16));
            hipLaunchKernelGGL(f.k, dim3(300), dim3(f.threads), 0, s, reinterpret_cast<const f32x4 *>(A), reinterpret_cast<f32x4 *>(B), vn);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(back.data(), B, vn * 16, hipMemcpyDeviceToHost));
            if (memcmp(pat.data(), back.data(), vn * 16) != 0) {
                fprintf(stderr, "form %s does not copy its input\n", f.name);
                return 3;
            }
        }
        CK(hipMemset(A, 0, vn * 16));
        CK(hipMemset(B, 0, vn * 16));
    }
    printf("{\"device\": \"%s\", \"cus\": %d, \"reps\": %d, \"trials\": %d, \"peak_tbps\": 8.0, \"rows\": [\n", prop.name, prop.multiProcessorCount, reps, trials);
    bool first = true;
    for (const Form &f : forms)
        for (double mb : moved) {
            const int streams = f.reads + f.writes;
            size_t n = (size_t)(mb / (16.0 * streams));
            n = n / 64 * 64;
            const size_t slice = ((n * 16 + 4095) / 4096) * 4096;
            const size_t slices = std::max<size_t>(std::min<size_t>(arena / slice, 64), 1);
            for (int g : wgs) {
                if (f.threads == 1024 && g > 512) // (16-wave workgroups: one or two per CU)
                    continue;
                std::vector<float> us;
                for (int t = 0; t < trials + 1; t++) {
                    CK(hipEventRecord(e0, s));
                    for (int r = 0; r < reps; r++) {
                        const size_t o = ((size_t)(t * reps + r) % slices) * slice;
                        hipLaunchKernelGGL(f.k, dim3(g), dim3(f.threads), 0, s, reinterpret_cast<const f32x4 *>(A + o), reinterpret_cast<f32x4 *>(B + o), n);
                    }
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (t > 0) // (the first trial warms the code object up)
                        us.push_back(ms * 1000.0f / reps);
                }
                std::sort(us.begin(), us.end());
                const double bytes = (double)n * 16.0 * streams;
                const double best = us.front(), med = us[us.size() / 2];
                printf("%s {\"form\": \"%s\", \"moved_mb\": %.1f, \"wgs\": %d, \"us_best\": %.2f, \"us_median\": %.2f, \"tbps_best\": %.3f, \"tbps_median\": %.3f, \"frac_of_8\": %.3f, \"rotates_over_mb\": %.0f}",
                       first ? " " : ",", f.name, bytes / 1e6, g, best, med, bytes / best / 1e6, bytes / med / 1e6, bytes / med / 1e6 / 8.0,
                       (double)slices * slice * (f.reads && f.writes ? 2 : 1) / 1e6);
                printf("\n");
                fflush(stdout);
                first = false;
            }
        }
    printf("]}\n");
    return 0;
}
