// Microbenchmark: do fp32 MFMA and ordinary VALU work overlap on a CDNA4 SIMD?
//   mode 0: MFMA waves only   mode 1: VALU waves only   mode 2: both, different waves of the same SIMD
//   mode 3: one wave issuing both, interleaved (independent)
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_mfma_f32_16x16x4_f32   1: v_mfma_f32_32x32x2_f32   2: v_mfma_f32_16x16x32_bf16
template <int KIND> struct Mma;
template <> struct Mma<0> { typedef f32x4 acc_t; static __device__ acc_t go(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); } };
template <> struct Mma<1> { typedef f32x16 acc_t; static __device__ acc_t go(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); } };
template <> struct Mma<2> { typedef f32x4 acc_t; static __device__ acc_t go(float a, float b, acc_t c) { bf16x8 x, y; for (int i = 0; i < 8; i++) { x[i] = (__bf16)a; y[i] = (__bf16)b; } return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0); } };

template <int KIND>
__global__ __launch_bounds__(512) void k(int mode, int iters, float *out)
{
    const int wave = threadIdx.x >> 6;
    typename Mma<KIND>::acc_t acc[4];
    for (int u = 0; u < 4; u++)
        for (int i = 0; i < (int)(sizeof(acc[0]) / 4); i++)
            acc[u][i] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8];
    for (int i = 0; i < 8; i++)
        v[i] = a + i;
    const bool do_mfma = (mode == 0 || mode == 2) ? wave < 4 : (mode == 3);
    const bool do_valu = (mode == 1 || mode == 2) ? wave >= 4 : (mode == 3);
    if (mode == 3 && wave >= 4)
        return;
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                acc[u] = Mma<KIND>::go(a, b, acc[u]);
#pragma unroll
                for (int i = 0; i < 7; i++) // 7 independent VALU ops per MFMA (8 issue slots of 4 cycles per 32-cycle MFMA)
                    v[i] = v[i] * b + a;
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int u = 0; u < 4; u++)
                acc[u] = Mma<KIND>::go(a, b, acc[u]);
    } else if (do_valu) {
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 7; i++)
                    v[i] = v[i] * b + a;
    }
    float s = 0;
    for (int u = 0; u < 4; u++)
        s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    for (int i = 0; i < 8; i++)
        s += v[i];
    if (s == 12345.678f)
        out[threadIdx.x] = s;
}

int main()
{
    float *out;
    hipMalloc(&out, 4096);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[4] = {"MFMA waves only (4 MFMA / iter / wave)", "VALU waves only (28 FMA / iter / wave)",
                            "both, different waves of one SIMD", "one wave, MFMA and VALU interleaved"};
    const char *kinds[3] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x32_bf16"};
    for (int kind = 0; kind < 3; kind++)
    for (int mode = 0; mode < 4; mode++) {
        auto kern = kind == 0 ? k<0> : (kind == 1 ? k<1> : k<2>);
        if (mode == 0)
            printf("%s\n", kinds[kind]);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, mode, 100, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d  %-45s %8.3f ms  (%.1f ns / iteration)\n", mode, names[mode], ms, ms * 1e6 / iters);
    }
    return 0;
}
