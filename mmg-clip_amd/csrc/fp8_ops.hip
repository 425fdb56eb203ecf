// OCP e4m3 operand preparation for the fp8 GEMM path (BASELINE config C5): per-tensor power-of-two weight scaling.
// Activations are cast unscaled by their producers (LayerNorm / GELU epilogues): e4m3 spans 2^-9 ... 448, which holds a
// LayerNorm output by construction and a GELU output with saturation; see DESIGN.md "fp8".
#include "common.h"

// amax[0] = max(amax[0], max |src|): non-negative floats order like their bit patterns, so an integer atomicMax does it
__global__ __launch_bounds__(256) void absmax_f32_kernel(const float* __restrict__ src, size_t n, unsigned* __restrict__ amax) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(src[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(m));
}

// scale = 2^floor(log2(448 / amax)) (1 when amax is 0 or not finite); dst = e4m3(src * scale); scales[0] = scale,
// scales[1] = 1 / scale.  Every workgroup derives the same scale from the device-resident amax: no host round trip.
__global__ __launch_bounds__(256) void quantize_e4m3_kernel(const float* __restrict__ src, size_t n4, size_t n,
                                                            const float* __restrict__ amax, unsigned* __restrict__ dst,
                                                            float* __restrict__ scales) {
    const float a = amax ? *amax : 0.f;
    float scale = 1.f;
    if (a > 0.f && a < 3.0e38f) scale = exp2f(floorf(log2f(448.f / a)));
    if (a * scale > 448.f) scale *= 0.5f;            // log2f rounding at an exact power of two
    if (blockIdx.x == 0 && threadIdx.x == 0 && scales) { scales[0] = scale; scales[1] = 1.f / scale; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (4 * i + e < n) ? src[4 * i + e] * scale : 0.f;
        dst[i] = pack4_e4m3(v[0], v[1], v[2], v[3]);
    }
}

MMG_API int mmg_absmax_f32(const float* src, long long n, float* amax, hipStream_t stream) {
    MMG_CHECK_ARG(src && amax && n > 0, "mmg_absmax_f32: bad argument");
    int blocks = cdiv(n, 256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(absmax_f32_kernel, dim3(blocks), dim3(256), 0, stream, src, (size_t)n, reinterpret_cast<unsigned*>(amax));
    MMG_LAUNCH_CHECK("mmg_absmax_f32");
    return 0;
}

MMG_API int mmg_quantize_e4m3_f32(const float* src, long long n, const float* amax, void* dst, float* scales, hipStream_t stream) {
    MMG_CHECK_ARG(src && dst && n > 0 && n % 4 == 0, "mmg_quantize_e4m3_f32: n=%lld must be a positive multiple of 4", n);
    const size_t n4 = (size_t)n / 4;
    int blocks = cdiv((long)n4, 256 * 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(quantize_e4m3_kernel, dim3(blocks), dim3(256), 0, stream, src, n4, (size_t)n, amax,
                       reinterpret_cast<unsigned*>(dst), scales);
    MMG_LAUNCH_CHECK("mmg_quantize_e4m3_f32");
    return 0;
}
