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
