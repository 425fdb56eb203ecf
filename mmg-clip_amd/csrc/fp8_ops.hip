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

// ---- gradients: bf16 -> OCP e5m2 with a per-tensor power-of-two scale (fp8 backward of config C5, round 4) -------------------------------------------
// amax[0] = max(amax[0], max |src|) over bf16 elements (16 bytes per lane)
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const uint4* __restrict__ src, size_t n8, unsigned* __restrict__ amax) {
    unsigned m = 0;                                           // |x| as bf16 bits in the high half: orders like the value
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned lo = (w[e] << 16) & 0x7fff0000u, hi = w[e] & 0x7fff0000u;
            m = lo > m ? lo : m; m = hi > m ? hi : m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(amax, m);          // (float bits; NaN / inf inputs give a NaN / inf amax: scale 1 below)
}

// scale = 2^floor(log2(16384 / amax)): the largest |src * scale| lands in [8192, 16384] - a factor 3.5 below e5m2's 57344, the headroom the
// data-gradient GEMM's output (written with the SAME scale) gets; 1 when amax is 0 or not finite.  dst = e5m2(src * scale), scales = (scale, 1 / scale).
__global__ __launch_bounds__(256) void quantize_e5m2_kernel(const uint4* __restrict__ src, size_t n8, const float* __restrict__ amax,
                                                            uint2* __restrict__ dst, float* __restrict__ scales) {
    const float a = *amax;
    float scale = 1.f;
    if (a > 0.f && a < 3.0e38f) scale = exp2f(floorf(log2f(16384.f / a)));
    if (a * scale > 16384.f) scale *= 0.5f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && scales) { scales[0] = scale; scales[1] = 1.f / scale; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        dst[i] = make_uint2(pack4_e5m2(bf2f_lo(v.x) * scale, bf2f_hi(v.x) * scale, bf2f_lo(v.y) * scale, bf2f_hi(v.y) * scale),
                            pack4_e5m2(bf2f_lo(v.z) * scale, bf2f_hi(v.z) * scale, bf2f_lo(v.w) * scale, bf2f_hi(v.w) * scale));
    }
}

// src bf16 [n] (n % 8 == 0, 16-byte aligned) -> dst e5m2 bytes [n], scales fp32 [2] = (scale, 1 / scale); amax fp32 [1] is scratch (zeroed here).
MMG_API int mmg_quantize_e5m2_bf16(const void* src, long long n, float* amax, void* dst, float* scales, hipStream_t stream) {
    MMG_CHECK_ARG(src && dst && amax && scales && n > 0 && n % 8 == 0, "mmg_quantize_e5m2_bf16: n=%lld must be a positive multiple of 8", n);
    const size_t n8 = (size_t)n / 8;
    int blocks = cdiv((long)n8, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (hipMemsetAsync(amax, 0, 4, stream) != hipSuccess) { mmg_set_error("mmg_quantize_e5m2_bf16: memset failed"); return 2; }
    hipLaunchKernelGGL(absmax_bf16_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, n8, reinterpret_cast<unsigned*>(amax));
    hipLaunchKernelGGL(quantize_e5m2_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, n8, amax, (uint2*)dst, scales);
    MMG_LAUNCH_CHECK("mmg_quantize_e5m2_bf16");
    return 0;
}

// Delayed scaling (one pass instead of two): the scale comes from the absmax this tensor had at its PREVIOUS quantisation (state[0]; the same block's
// gradient of the previous micro-batch / step - gradient magnitudes move slowly), with 14 x headroom (target 4096 instead of 16384: e5m2's range is
// 2^32, the low end has the room), and the pass records the new absmax in state[1] for the next call (the host swaps the two).  state[0] = 0: scale 1.
__global__ __launch_bounds__(256) void quantize_e5m2_delayed_kernel(const uint4* __restrict__ src, size_t n8, const float* __restrict__ amax_prev,
                                                                    unsigned* __restrict__ amax_next, uint2* __restrict__ dst, float* __restrict__ scales) {
    const float a = *amax_prev;
    float scale = 1.f;
    if (a > 0.f && a < 3.0e38f) scale = exp2f(floorf(log2f(4096.f / a)));
    if (a * scale > 4096.f) scale *= 0.5f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && scales) { scales[0] = scale; scales[1] = 1.f / scale; }
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned lo = (w[e] << 16) & 0x7fff0000u, hi = w[e] & 0x7fff0000u;
            m = lo > m ? lo : m; m = hi > m ? hi : m;
        }
        dst[i] = make_uint2(pack4_e5m2(bf2f_lo(v.x) * scale, bf2f_hi(v.x) * scale, bf2f_lo(v.y) * scale, bf2f_hi(v.y) * scale),
                            pack4_e5m2(bf2f_lo(v.z) * scale, bf2f_hi(v.z) * scale, bf2f_lo(v.w) * scale, bf2f_hi(v.w) * scale));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(amax_next, m);
}

// As mmg_quantize_e5m2_bf16 with the scale taken from amax_prev (fp32 [1], device) and this tensor's own absmax left in amax_next (fp32 [1], zeroed here).
MMG_API int mmg_quantize_e5m2_bf16_delayed(const void* src, long long n, const float* amax_prev, float* amax_next, void* dst, float* scales,
                                           hipStream_t stream) {
    MMG_CHECK_ARG(src && dst && amax_prev && amax_next && scales && n > 0 && n % 8 == 0, "mmg_quantize_e5m2_bf16_delayed: n=%lld must be a positive multiple of 8", n);
    const size_t n8 = (size_t)n / 8;
    int blocks = cdiv((long)n8, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (hipMemsetAsync(amax_next, 0, 4, stream) != hipSuccess) { mmg_set_error("mmg_quantize_e5m2_bf16_delayed: memset failed"); return 2; }
    hipLaunchKernelGGL(quantize_e5m2_delayed_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, n8, amax_prev,
                       reinterpret_cast<unsigned*>(amax_next), (uint2*)dst, scales);
    MMG_LAUNCH_CHECK("mmg_quantize_e5m2_bf16_delayed");
    return 0;
}

// Delayed-scaling cast of a gradient matrix AND its column sums in one pass (round 4): the incoming gradient dy [M, C] of a CNBlock's MLP is read
// once - e5m2 bytes out (16-byte stores: a thread owns 16 columns), its absmax for the next call, and colsum[c] += sum_m dy[m][c] (the bias
// gradient of the block's second Linear, which the 8-bit weight-gradient GEMM takes from the bf16 gradient, not from its cast).  Replaces
// mmg_quantize_e5m2_bf16_delayed + mmg_colsum_bf16 (two reads of dy).  C % 16 == 0 and 256 % (C / 16) == 0: a thread keeps its columns for all its rows.
__global__ __launch_bounds__(256) void quantize_e5m2_colsum_kernel(const uint4* __restrict__ src, int M, int C, const float* __restrict__ amax_prev,
                                                                   unsigned* __restrict__ amax_next, uint4* __restrict__ dst,
                                                                   float* __restrict__ scales, float* __restrict__ colsum) {
    extern __shared__ float q5_sums[];          // [C]
    const float a = *amax_prev;
    float scale = 1.f;
    if (a > 0.f && a < 3.0e38f) scale = exp2f(floorf(log2f(4096.f / a)));
    if (a * scale > 4096.f) scale *= 0.5f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && scales) { scales[0] = scale; scales[1] = 1.f / scale; }
    for (int i = threadIdx.x; i < C; i += 256) q5_sums[i] = 0.f;
    __syncthreads();
    const int G = C / 16;                        // threads per row
    const int cg = threadIdx.x % G;
    const int rows_wg = 256 / G;                 // rows a workgroup covers per pass
    const long stride = (long)gridDim.x * rows_wg;
    float cs[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[e] = 0.f;
    unsigned m = 0;
    auto one = [&](long r, const uint4 v0, const uint4 v1) {
        const unsigned w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f0 = bf2f_lo(w[2 * e]), f1 = bf2f_hi(w[2 * e]), f2 = bf2f_lo(w[2 * e + 1]), f3 = bf2f_hi(w[2 * e + 1]);
            cs[4 * e] += f0; cs[4 * e + 1] += f1; cs[4 * e + 2] += f2; cs[4 * e + 3] += f3;
            o[e] = pack4_e5m2(f0 * scale, f1 * scale, f2 * scale, f3 * scale);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned lo = (w[e] << 16) & 0x7fff0000u, hi = w[e] & 0x7fff0000u;
            m = lo > m ? lo : m; m = hi > m ? hi : m;
        }
        dst[r * G + cg] = make_uint4(o[0], o[1], o[2], o[3]);
    };
    long r = (long)blockIdx.x * rows_wg + threadIdx.x / G;
    for (; r + stride < M; r += 2 * stride) {                      // two rows (four 16-byte loads) in flight per thread
        const uint4 a0 = src[(r * G + cg) * 2], a1 = src[(r * G + cg) * 2 + 1];
        const uint4 b0 = src[((r + stride) * G + cg) * 2], b1 = src[((r + stride) * G + cg) * 2 + 1];
        one(r, a0, a1);
        one(r + stride, b0, b1);
    }
    if (r < M) one(r, src[(r * G + cg) * 2], src[(r * G + cg) * 2 + 1]);
#pragma unroll
    for (int e = 0; e < 16; ++e) atomicAdd(&q5_sums[cg * 16 + e], cs[e]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(amax_next, m);
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) atomicAdd(colsum + i, q5_sums[i]);
}

MMG_API int mmg_quantize_e5m2_colsum_bf16(const void* src, int M, int C, const float* amax_prev, float* amax_next, void* dst, float* scales,
                                          float* colsum, hipStream_t stream) {
    MMG_CHECK_ARG(src && dst && amax_prev && amax_next && scales && colsum && M > 0, "mmg_quantize_e5m2_colsum_bf16: null pointer or M=%d", M);
    MMG_CHECK_ARG(C >= 16 && C % 16 == 0 && C <= 4096 && 256 % (C / 16) == 0,
                  "mmg_quantize_e5m2_colsum_bf16: C=%d must be 16 x a divisor of 256 (16 ... 4096)", C);
    const int rows_wg = 256 / (C / 16);
    int blocks = cdiv(M, rows_wg * 8);            // >= 8 rows per thread
    const int cap = 4 * mmg_cu_count_cached();
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (hipMemsetAsync(amax_next, 0, 4, stream) != hipSuccess) { mmg_set_error("mmg_quantize_e5m2_colsum_bf16: memset failed"); return 2; }
    hipLaunchKernelGGL(quantize_e5m2_colsum_kernel, dim3(blocks), dim3(256), C * sizeof(float), stream, (const uint4*)src, M, C, amax_prev,
                       reinterpret_cast<unsigned*>(amax_next), (uint4*)dst, scales, colsum);
    MMG_LAUNCH_CHECK("mmg_quantize_e5m2_colsum_bf16");
    return 0;
}
