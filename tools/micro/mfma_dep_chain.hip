// Microbenchmark: cost of back-to-back DEPENDENT fp32 MFMAs (same accumulator) vs interleaved independent ones.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_dep_chain mfma_dep_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, float *out)
{
    f32x4 acc[NACC];
    for (int u = 0; u < NACC; u++)
        acc[u] = (f32x4){0, 0, 0, 0};
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int j = 0; j < 12 / NACC; j++)
#pragma unroll
            for (int u = 0; u < NACC; u++)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
    float s = 0;
    for (int u = 0; u < NACC; u++)
        s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    if (s == 12345.678f)
        out[threadIdx.x] = s;
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 4096);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int nacc : {1, 2, 3, 4}) {
        for (int waves = 1; waves <= 2; waves++) { // waves per SIMD
            auto kern = nacc == 1 ? k<1> : nacc == 2 ? k<2> : nacc == 3 ? k<3> : k<4>;
            hipLaunchKernelGGL(kern, dim3(256 * waves), dim3(256), 0, 0, 100, out);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256 * waves), dim3(256), 0, 0, iters, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%d accumulator(s) round-robin, %d wave(s)/SIMD: %.2f ns per MFMA per SIMD\n", nacc, waves,
                   ms * 1e6 / iters / 12 / waves);
        }
    }
    return 0;
}
