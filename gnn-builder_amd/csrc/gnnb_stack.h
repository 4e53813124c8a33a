// gnnb_stack.h -- pieces shared by the LDS-resident conv-stack kernels (k_stack.hip: k_gcn2_fused; k_stack_zf.hip: k_gcn2_zf)
#pragma once
#include "gnnb_device.h"

namespace gnnb {

__host__ __device__ constexpr int g2_units(int math) { return math ? 3 : 4; }
static_assert(16 * g2_units(0) == GNNB_G2_STAGE_ROWS && 16 * g2_units(1) == GNNB_G2_STAGE_ROWS_BF6, "graph prep picks the tile size against these");
static constexpr int G2_TCAP = 64;           // tile-table entries a workgroup keeps in LDS
static constexpr int G2_WG = 512;            // 8 waves; two workgroups per CU = 4 waves per SIMD
static constexpr int G2_NW = G2_WG / 64;

struct G2Stage {
    int ta, tb, nb, rows, ga, gb;
};

// Sum / max of a value over the four 16-lane rows of a wave (same lane index in each row) with the
// gfx950 row-swap instructions -- two VALU operations per step instead of an LDS crossbar round trip.
__device__ __forceinline__ float rows4_sum(float x)
{
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows4_max(float x)
{
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// Workgroup barrier of the fused kernel: LDS traffic drained, NO vector-memory drain.  __syncthreads()
// carries a fence, for which the compiler emits s_waitcnt vmcnt(0) whenever it has stores of its own in
// flight (the pooled outputs) -- and that would also wait for the untracked DMA of the next stage.
__device__ __forceinline__ void g2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


} // namespace gnnb
