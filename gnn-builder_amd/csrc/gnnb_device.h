// gnnb_device.h -- device-side helpers shared by the kernel translation units of libgnnb_hip.so (gfx950 only):
// dynamic-LDS attribute cache, LDS-DMA issue / counted-wait helpers, vector-of-floats helpers, activations.
// Everything here is static / inline: each translation unit gets its own copy (the library is built without -fgpu-rdc).
#pragma once
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "gnnb_internal.h"

namespace gnnb {

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised first.
// The attribute is per device and per kernel: remembered here per (device, kernel) under a lock, so that launches
// from several host threads or on several devices of one process each get it, and the runtime call (a few
// microseconds of host time) is not paid on every launch.
static hipError_t ensure_dynamic_lds(const void *kern, size_t lds)
{
    if (lds <= 64 * 1024)
        return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[std::make_pair(dev, kern)];
    if (have >= lds)
        return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
        have = lds;
    return e;
}

// CUs of the CURRENT device (cached per device under a lock: a process may drive several GPUs)
static int device_cu_count()
{
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end())
        return it->second;
    hipDeviceProp_t prop;
    const int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    cus[dev] = n;
    return n;
}

static constexpr int WG = 256; // 4 wavefronts
// wave priority of the small kernels that co-run with another batch's conv-stack kernel (graph prep, readout, k_conv_rows)
#ifndef GNNB_GUEST_PRIO
#define GNNB_GUEST_PRIO 3
#endif

// compile-time integer tag (generic lambdas dispatch on it)
template <int V>
struct IntTag {
    static constexpr int value = V;
};

// Diagnostic build only (-DGNNB_PROBE, tools/probe_agg.py): per-workgroup phase stamps.  The
// product library is built without it and executes no stamp.
#ifdef GNNB_PROBE
__device__ unsigned long long g_probe[16 * 8192];
#define GNNB_STAMP(slot)                                                                   \
    do {                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 8192) {                                       \
            g_probe[blockIdx.x * 8 + 2 * (slot)] = wall_clock64();                         \
            g_probe[blockIdx.x * 8 + 2 * (slot) + 1] = clock64();                          \
        }                                                                                  \
    } while (0)
#define GNNB_STAMP_END(slot)                                                               \
    do {                                                                                   \
        __builtin_amdgcn_s_waitcnt(0); /* drain this wave's stores first */                \
        GNNB_STAMP(slot);                                                                  \
    } while (0)
#else
#define GNNB_STAMP(slot) do { } while (0)
#define GNNB_STAMP_END(slot) do { } while (0)
#endif

template <int VEC>
struct Vf;
template <>
struct Vf<4> {
    float4 v;
    __device__ static Vf load(const float *p) { Vf r; r.v = *reinterpret_cast<const float4 *>(p); return r; }
    __device__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = v; }
    __device__ static Vf splat(float s) { Vf r; r.v = make_float4(s, s, s, s); return r; }
};
template <>
struct Vf<1> {
    float v;
    __device__ static Vf load(const float *p) { Vf r; r.v = *p; return r; }
    __device__ void store(float *p) const { *p = v; }
    __device__ static Vf splat(float s) { Vf r; r.v = s; return r; }
};
#define VF_BINOP(NAME, EXPR)                                                         \
    __device__ inline Vf<4> NAME(const Vf<4> &a, const Vf<4> &b)                     \
    {                                                                                \
        Vf<4> r;                                                                     \
        { const float x = a.v.x, y = b.v.x; r.v.x = (EXPR); }                        \
        { const float x = a.v.y, y = b.v.y; r.v.y = (EXPR); }                        \
        { const float x = a.v.z, y = b.v.z; r.v.z = (EXPR); }                        \
        { const float x = a.v.w, y = b.v.w; r.v.w = (EXPR); }                        \
        return r;                                                                    \
    }                                                                                \
    __device__ inline Vf<1> NAME(const Vf<1> &a, const Vf<1> &b)                     \
    {                                                                                \
        Vf<1> r;                                                                     \
        const float x = a.v, y = b.v;                                                \
        r.v = (EXPR);                                                                \
        return r;                                                                    \
    }
VF_BINOP(vadd, x + y)
VF_BINOP(vmul, x *y)
VF_BINOP(vmax, fmaxf(x, y))
VF_BINOP(vmin, fminf(x, y))
VF_BINOP(vdiv, x / y)
VF_BINOP(vsub, x - y)
#undef VF_BINOP
// PyG StdAggregation: var = E[h^2] - E[h]^2 ; std = sqrt(clamp(var, 1e-5)) ; 0 where <= sqrt(1e-5)
__device__ inline float pyg_std1(float mean2, float mean)
{
    float var = mean2 - mean * mean;
    var = var < 1e-5f ? 1e-5f : var;
    const float sd = sqrtf(var);
    return sd <= sqrtf(1e-5f) ? 0.0f : sd;
}
__device__ inline Vf<4> pyg_std(const Vf<4> &m2, const Vf<4> &m)
{
    Vf<4> r;
    r.v = make_float4(pyg_std1(m2.v.x, m.v.x), pyg_std1(m2.v.y, m.v.y), pyg_std1(m2.v.z, m.v.z),
                      pyg_std1(m2.v.w, m.v.w));
    return r;
}
__device__ inline Vf<1> pyg_std(const Vf<1> &m2, const Vf<1> &m)
{
    Vf<1> r;
    r.v = pyg_std1(m2.v, m.v);
    return r;
}

// -------------------------------------------------------------------------------------
// ---- Overflow contract of the REDUCED-precision math modes (gnnb_model_desc::math 2 "bf16x3" / 3 "f16x3"; VERDICT round 5,
// item 3).  The reference's fixed-point build has DEFINED overflow (ap_fixed<W, I, AP_TRN, AP_WRAP>, code_gen.py:39-52,
// model.h.jinja:38-62); fp16 pieces do not: an activation or weight of 65504 and above becomes inf, silently.  Every kernel
// that multiplies on 16-bit pieces therefore looks at what it PRODUCED (the fp32 accumulators of its reduced products): a
// non-finite value sets bit GNNB_FLAG_RANGE of the workspace's flag word (BatchTables::err, beside graph prep's bits) and
// gnnb_workspace_check returns GNNB_ERR_RANGE -- the caller reruns the model with math = 0.  Cost: one v_cmp_class per
// accumulator value and one scalar OR, in the opt-in modes only; the fp32 instantiations carry none of it.
struct RangeProbe {
    // (a wave-uniform lane mask, i.e. a scalar register pair: as a per-lane flag carried across a stage's barriers it became a
    // vector register in the deep-GCN stack variants that sit at the 128-register budget, and spilled)
    unsigned long long bad = 0ull;
    __device__ __forceinline__ void see(float v, bool in_range = true)
    {
        bad |= __ballot(in_range && __builtin_amdgcn_classf(v, 0x207)); // sNaN | qNaN | -inf | +inf
    }
    template <typename ACC, int N>
    __device__ __forceinline__ void see_vec(const ACC &a, bool in_range = true)
    {
#pragma unroll
        for (int i = 0; i < N; i++)
            see(a[i], in_range);
    }
    __device__ __forceinline__ bool any() const { return bad != 0ull; } // wave-uniform
    // one atomic per wave that saw something (rare), and the host-mapped copy graph prep's flag_batch writes too
    // returns the number of vector-memory instructions the WAVE issued (wave-uniform: 0, or the atomic + the host-mapped store):
    // the kernels that count their own vmcnt add it to their books
    __device__ __forceinline__ int report(int32_t *err, int32_t *err_host) const
    {
        if (!err || bad == 0ull)
            return 0;
        if ((threadIdx.x & 63) == 0) {
            atomicOr(err, GNNB_FLAG_RANGE);
            if (err_host)
                *reinterpret_cast<volatile int32_t *>(err_host) = GNNB_FLAG_RANGE;
        }
        return err_host ? 2 : 1;
    }
};

// The run [c0, c1) of `total` units that workgroup b of G takes: floor(b total / G) .. floor((b + 1) total / G), in 32-bit
// arithmetic.  Written as (long long) b * total / G the two cuts were two SOFTWARE 64-bit divisions -- ~300 scalar instructions
// per wave in front of every persistent kernel's first DMA (round 6, read from the ISA: 1.1 M of k_gcn2_zf's 3.7 M scalar
// instructions per launch at BASELINE config 2 were these).  Exact: total = q G + r  =>  floor(b total / G) = b q + floor(b r / G),
// and b r < G^2 fits 32 bits for grids below 65536 workgroups (the launchers' grids are a few per CU).
__device__ __forceinline__ void run_cuts(unsigned b, unsigned G, unsigned total, int &c0, int &c1)
{
    const unsigned q = total / G, r = total - q * G;
    c0 = (int)(b * q + (b * r) / G);
    c1 = (int)((b + 1) * q + ((b + 1) * r) / G);
}

// LDS-DMA helpers (global_load_lds: global -> LDS without VGPR staging).
typedef __attribute__((address_space(3))) void *lds_vptr;
typedef const __attribute__((address_space(1))) void *glb_vptr;

// LDS destination = wave-uniform base + lane * size (cdna_hip_programming.md section 5, Caveat);
// the size argument must be a literal, so one function per width.
__device__ inline void dma16_to_lds(const void *gsrc_lane, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((glb_vptr)gsrc_lane, (lds_vptr)lds_wave_base, 16, 0, 0);
}
__device__ inline void dma4_to_lds(const void *gsrc_lane, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((glb_vptr)gsrc_lane, (lds_vptr)lds_wave_base, 4, 0, 0);
}

// copy `count` dwords global -> LDS, spread over the workgroup's waves, 64 dwords per instruction
__device__ inline void dma_dwords(const void *g, void *l, int count, int wave, int lane, int nwaves)
{
    const char *gs = reinterpret_cast<const char *>(g);
    char *ls = reinterpret_cast<char *>(l);
    for (int c = wave * 64; c < count; c += nwaves * 64)
        if (c + lane < count)
            dma4_to_lds(gs + (size_t)(c + lane) * 4, ls + (size_t)c * 4);
}

// "Untracked" forms for software-pipelined kernels.  The compiler's waitcnt pass cannot tell which
// LDS bytes an in-flight LDS-DMA will write (dynamic shared memory carries no alias scopes), so after
// the builtin it puts s_waitcnt vmcnt(0) in front of EVERY later ds_read -- which serialises "issue
// the next stage's DMA, then compute on the current stage" completely.  Issued from inline assembly
// the DMA is invisible to that pass; the kernel then owns the ordering and MUST wait itself
// (dma_wait_all / a counted s_waitcnt, then a barrier) before any wave reads the destination.
// Compiler-inserted vmcnt waits stay correct: extra outstanding operations only make vmcnt(N) stronger.
__device__ inline void dma16_to_lds_u(const void *gsrc_lane, void *lds_wave_base)
{
    const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_vptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(a), "v"(gsrc_lane) : "memory");
}
// scalar-base form: address = wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset; the LDS
// destination is a wave-uniform LDS byte address.  Keeps a streaming kernel's per-chunk address arithmetic on the
// scalar unit (fp32 MFMA and VALU instructions share one issue port).
__device__ inline void dma16_to_lds_s(const void *gbase_uniform, uint32_t lane_byte_off, uint32_t lds_addr_uniform)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr_uniform), "v"(lane_byte_off),
                 "s"(gbase_uniform)
                 : "memory");
}
__device__ inline void dma4_to_lds_u(const void *gsrc_lane, void *lds_wave_base)
{
    const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_vptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(a), "v"(gsrc_lane) : "memory");
}
__device__ inline void dma_dwords_u(const void *g, void *l, int count, int wave, int lane, int nwaves)
{
    const char *gs = reinterpret_cast<const char *>(g);
    char *ls = reinterpret_cast<char *>(l);
    for (int c = wave * 64; c < count; c += nwaves * 64)
        if (c + lane < count)
            dma4_to_lds_u(gs + (size_t)(c + lane) * 4, ls + (size_t)c * 4);
}
// Inclusive prefix sum over the 64 lanes of a wave with DPP adds: Hillis-Steele inside each row of 16 lanes (row_shr
// 1, 2, 4, 8; lanes without a source add 0), then the row totals (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2
// and 3).  Six vector instructions; the __shfl_up form is six ds_bpermute round trips with ~7 instructions each.
__device__ __forceinline__ int wave_scan_incl(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ inline void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n: the instruction takes an immediate
__device__ __forceinline__ void vmcnt_wait_upto(int n)
{
    switch (__builtin_amdgcn_readfirstlane(n)) { // scalar branch

#define GNNB_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    GNNB_VMW(1) GNNB_VMW(2) GNNB_VMW(3) GNNB_VMW(4) GNNB_VMW(5) GNNB_VMW(6) GNNB_VMW(7) GNNB_VMW(8) GNNB_VMW(9)
    GNNB_VMW(10) GNNB_VMW(11) GNNB_VMW(12) GNNB_VMW(13) GNNB_VMW(14) GNNB_VMW(15) GNNB_VMW(16) GNNB_VMW(17) GNNB_VMW(18)
#undef GNNB_VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}


// vmcnt wait with a run-time, wave-uniform count (the instruction takes an immediate): 0..63
__device__ __forceinline__ void vmcnt_wait_n(int n)
{
    switch (__builtin_amdgcn_readfirstlane(n)) { // scalar jump table
#define GNNB_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define GNNB_VMW8(b) GNNB_VMW(b) GNNB_VMW(b + 1) GNNB_VMW(b + 2) GNNB_VMW(b + 3) GNNB_VMW(b + 4) GNNB_VMW(b + 5) GNNB_VMW(b + 6) GNNB_VMW(b + 7)
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    GNNB_VMW(1) GNNB_VMW(2) GNNB_VMW(3) GNNB_VMW(4) GNNB_VMW(5) GNNB_VMW(6) GNNB_VMW(7)
    GNNB_VMW(8) GNNB_VMW(9) GNNB_VMW(10) GNNB_VMW(11) GNNB_VMW(12) GNNB_VMW(13) GNNB_VMW(14) GNNB_VMW(15)
    GNNB_VMW(16) GNNB_VMW(17) GNNB_VMW(18) GNNB_VMW(19) GNNB_VMW(20) GNNB_VMW(21) GNNB_VMW(22) GNNB_VMW(23)
    GNNB_VMW(24) GNNB_VMW(25) GNNB_VMW(26) GNNB_VMW(27) GNNB_VMW(28) GNNB_VMW(29) GNNB_VMW(30) GNNB_VMW(31)
    GNNB_VMW(32) GNNB_VMW(33) GNNB_VMW(34) GNNB_VMW(35) GNNB_VMW(36) GNNB_VMW(37) GNNB_VMW(38) GNNB_VMW(39)
    GNNB_VMW(40) GNNB_VMW(41) GNNB_VMW(42) GNNB_VMW(43) GNNB_VMW(44) GNNB_VMW(45) GNNB_VMW(46) GNNB_VMW(47)
    GNNB_VMW(48) GNNB_VMW(49) GNNB_VMW(50) GNNB_VMW(51) GNNB_VMW(52) GNNB_VMW(53) GNNB_VMW(54) GNNB_VMW(55)
    GNNB_VMW(56) GNNB_VMW(57) GNNB_VMW(58) GNNB_VMW(59) GNNB_VMW(60) GNNB_VMW(61) GNNB_VMW(62)
#undef GNNB_VMW8
#undef GNNB_VMW
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}

// 16-B (or 4-B) row-piece store, optionally non-temporal (the output is not re-read by this kernel)
typedef float agg_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void agg_store(const Vf<4> &v, float *p)
{
    if (NT) {
        agg_f32x4 t = {v.v.x, v.v.y, v.v.z, v.v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<agg_f32x4 *>(p));
    } else {
        v.store(p);
    }
}
template <bool NT>
__device__ __forceinline__ void agg_store(const Vf<1> &v, float *p)
{
    if (NT)
        __builtin_nontemporal_store(v.v, p);
    else
        v.store(p);
}

// LGConv normaliser (gnn_builder_lib.h:2383-2386): 1/sqrt(d_i d_j) on IN-degrees, 0 when the product is 0
__device__ __forceinline__ float lg_coef(int di, int dj)
{
    const int pr = di * dj;
    return pr > 0 ? __frsqrt_rn((float)pr) : 0.0f; // (v_rsq_f32: 1 ulp; the oracle's 1/sqrt differs by < 2e-7 relative)
}

static constexpr int BM = 128;
static constexpr int BK = 32;
static constexpr int LDS_LD = BK + 4; // padded row, floats

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline float act_apply(float v, int act)
{
    switch (act) {
    case GNNB_ACT_RELU:
        return v > 0.0f ? v : 0.0f; // gnn_builder_lib.h:363-375
    case GNNB_ACT_GELU:
        return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); // nn.GELU (erf), lib:378-385
    case GNNB_ACT_SIGMOID:
        return 1.0f / (1.0f + expf(-v)); // lib:420-425
    case GNNB_ACT_TANH:
        return tanhf(v); // lib:436-448
    default:
        return v;
    }
}

// Compile-time activation: the epilogues dispatch on `act` ONCE and run a straight-line copy of
// the store loop per activation (a runtime switch inside the unrolled loops replicates the inlined
// erff/tanhf/expf bodies per element: thousands of instructions and hundreds of branches).
template <int ACT>
__device__ inline float act_t(float v)
{
    if (ACT == GNNB_ACT_RELU) {
        // ONE v_max_f32: written as `v > 0 ? v : 0` (or fmaxf) the compiler canonicalises an operand it cannot prove quiet
        // first -- v_max v, v, v in front of every v_max v, 0, v behind an MFMA: twice the instructions in every epilogue
        // (round 6, read from k_gcn2_zf's ISA; fp32 MFMA and VALU share the issue port).  Same result for every input but a
        // signalling NaN (quieted instead of 0), which no kernel produces.
        float r;
        asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
        return r;
    }
    if (ACT == GNNB_ACT_GELU)
        return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (ACT == GNNB_ACT_SIGMOID)
        return 1.0f / (1.0f + expf(-v));
    if (ACT == GNNB_ACT_TANH)
        return tanhf(v);
    return v;
}
// calls f(IntTag<act>{}) with `act` turned into a compile-time constant
#define GNNB_DISPATCH_ACT(act, f)                        \
    switch (act) {                                       \
    case GNNB_ACT_RELU: f(IntTag<GNNB_ACT_RELU>{}); break;       \
    case GNNB_ACT_GELU: f(IntTag<GNNB_ACT_GELU>{}); break;       \
    case GNNB_ACT_SIGMOID: f(IntTag<GNNB_ACT_SIGMOID>{}); break; \
    case GNNB_ACT_TANH: f(IntTag<GNNB_ACT_TANH>{}); break;       \
    default: f(IntTag<GNNB_ACT_NONE>{}); break;                  \
    }

__device__ inline float4 load4_guard(const float *p, int remaining, bool vec)
{
    // `remaining` = number of valid floats at p (<= 0: none)
    if (remaining <= 0)
        return make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && remaining >= 4)
        return *reinterpret_cast<const float4 *>(p);
    float4 r;
    r.x = p[0];
    r.y = remaining > 1 ? p[1] : 0.f;
    r.z = remaining > 2 ? p[2] : 0.f;
    r.w = remaining > 3 ? p[3] : 0.f;
    return r;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- fp32 product through the bf16 matrix cores ("bf16x6").  x = h + m + l EXACTLY, each piece a bf16
// (8 significant bits each: truncate, subtract, truncate, subtract -- every step is exact in fp32), so
// a.b = sum of nine bf16 x bf16 products, each exact in fp32.  The six with i + j <= 2 are kept
// (hh, hm, mh, hl, lh, mm); the three dropped ones are below 2^-24 |a||b|, i.e. below what fp32 resolves of
// the product.  Accumulation is fp32 inside v_mfma_f32_16x16x32_bf16.  Cost: 6 MFMA of 4 passes per
// 32-wide k block instead of 8 fp32 MFMA of 8 passes -- 2.4x fewer pipe cycles, and fp32 MFMA runs at
// the vector-FMA rate on this chip (tools/micro/mfma_valu_overlap.hip).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3(float x, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xffff0000u;
    l = __float_as_uint(r1 - __uint_as_float(m)); // <= 8 significant bits left: its upper half is exact
}
// two fp32 bit patterns -> their upper halves packed as {bf16(a) in bits 0..15, bf16(b) in bits 16..31}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v)
{
    union {
        u32x4 u;
        bf16x8 b;
    } c;
    c.u = v;
    return c.b;
}

// split 8 consecutive fp32 values (two float4) into the three bf16x8 pieces of an MFMA operand
__device__ __forceinline__ void split3x8(const float4 &f0, const float4 &f1, u32x4 &h, u32x4 &m, u32x4 &l)
{
    const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t h0, m0, l0, h1, m1, l1;
        split3(v[2 * i], h0, m0, l0);
        split3(v[2 * i + 1], h1, m1, l1);
        h[i] = pack_hi16(h0, h1);
        m[i] = pack_hi16(m0, m1);
        l[i] = pack_hi16(l0, l1);
    }
}

// ---- fp16 pieces ("f16x3"): x = hi + mid + e with hi = fp16(x), mid = fp16(x - hi) (round to nearest even, the difference is
// exact in fp32): 11 + 11 significant bits, |e| <= 2^-22 |x| -- but fp16's range (|x| < 65504; pieces below 6e-8 are lost)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_f16(_Float16 a, _Float16 b)
{
    union {
        f16x2 h;
        uint32_t u;
    } c;
    c.h = (f16x2){a, b};
    return c.u;
}
// eight consecutive fp32 values -> their hi and mid fp16 pieces as the 16-B operands of a x16 / x32 f16 MFMA
__device__ __forceinline__ void split2x8_f16(const float4 &f0, const float4 &f1, u32x4 &h, u32x4 &m)
{
    const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const _Float16 h0 = (_Float16)v[2 * i], h1 = (_Float16)v[2 * i + 1];
        h[i] = pack_f16(h0, h1);
        m[i] = pack_f16((_Float16)(v[2 * i] - (float)h0), (_Float16)(v[2 * i + 1] - (float)h1));
    }
}
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v)
{
    union {
        u32x4 u;
        f16x8 h;
    } c;
    c.u = v;
    return c.h;
}


} // namespace gnnb
