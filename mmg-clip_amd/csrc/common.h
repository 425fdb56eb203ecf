// Shared device/host helpers for the mmg-clip gfx950 kernels.
// Everything here is written for CDNA4 (wave64, MFMA, 160 KiB LDS); there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define MMG_API extern "C" __attribute__((visibility("default")))

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) short bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;  // raw bf16 bits

// ---- host-side error plumbing (C ABI: int return + mmg_last_error()) ----
void mmg_set_error(const char* fmt, ...);
#define MMG_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            mmg_set_error(__VA_ARGS__);          \
            return 1;                            \
        }                                        \
    } while (0)
#define MMG_LAUNCH_CHECK(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            mmg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

// ---- bf16 <-> f32 ----
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ float bf2f_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf2f_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2v;
    bf2v v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

// ---- wave64 reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-GELU, as torch.nn.GELU() / HF "gelu".  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16
// resolution of the tensors it is applied to): one v_exp + one v_rcp + 6 FMAs instead of libm erff's ~40 instructions.
// It sits in GEMM epilogues, where the libm version cost more than the MFMA main loop.  Both helpers share the
// exp(-x^2/2) between the erf tail and the Gaussian density.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));     // v_rcp_f32, 1 ulp
    const float e = __expf(-ax * ax);                       // = exp(-x^2 / 2)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float tail = 0.5f * poly * t * e;                 // 0.5 * erfc(|x|/sqrt2)
    cdf = x >= 0.f ? 1.0f - tail : tail;
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return fmaf(x, pdf, cdf);
}

// 16-byte store that bypasses L2 allocation when `nt` (streaming outputs larger than the Infinity Cache)
__device__ __forceinline__ void store16_stream(void* dst, const uint4 v, bool nt) {
    if (nt) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(dst));
    } else {
        *reinterpret_cast<uint4*>(dst) = v;
    }
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// gfx950 has 160 KiB of LDS per CU; anything above 64 KiB of dynamic LDS must be opted into.
template <typename F>
static inline void mmg_allow_lds(F* f, size_t bytes) {
    (void)hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
