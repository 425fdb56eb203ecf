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
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;  // raw bf16 bits

// ---- host-side error plumbing (C ABI: int return + mmg_last_error()) ----
void mmg_set_error(const char* fmt, ...);
// diagnostics (core.hip): the dispatchers record the name of the kernel instantiation they launch while notes are on
int mmg_kernel_notes_on(void);
void mmg_note_kernel(const char* fmt, ...);
int mmg_cu_count_cached(void);        // CUs of the current device, cached per device id (core.hip)
#define MMG_NOTE_KERNEL(...) do { if (mmg_kernel_notes_on()) mmg_note_kernel(__VA_ARGS__); } while (0)
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
// four floats -> four OCP e4m3 bytes (gfx950 v_cvt_pk_fp8_f32, round-to-nearest-even), saturating at +-448
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

// four floats -> four OCP e5m2 bytes (v_cvt_pk_bf8_f32, round-to-nearest-even), saturating at +-57344
__device__ __forceinline__ unsigned pack4_e5m2(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -57344.f, 57344.f); b = __builtin_amdgcn_fmed3f(b, -57344.f, 57344.f);
    c = __builtin_amdgcn_fmed3f(c, -57344.f, 57344.f); d = __builtin_amdgcn_fmed3f(d, -57344.f, 57344.f);
    int v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
    return (unsigned)v;
}

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
// resolution of the tensors it is applied to).  It sits in GEMM epilogues and in the fused CNBlock MLP, where the VALU is
// the bound, so the form is chosen for instruction count (13 VALU ops, 2 of them transcendental, no compare/select):
//     s = |x| sqrt(log2(e)/2),  e = 2^(-s^2) = exp(-x^2/2),  t = 1/(1 + p |x|/sqrt2),  tail = 0.5 erfc(|x|/sqrt2) = q(t) t e
//     GELU(x)  = max(x,0) - |x| tail = x/2 + |x| (1/2 - tail)   (x Phi(x) with Phi = 1 - tail for x >= 0, tail for x < 0)
//     GELU'(x) = 0.5 + copysign(0.5 + w, x),  w = e (|x|/sqrt(2 pi) - q(t) t)      (= Phi(x) + x phi(x))
__device__ __forceinline__ void gelu_core(float x, float& ax, float& e, float& qt) {
    ax = fabsf(x);
    const float s = ax * 0.84932180028801904f;                             // sqrt(log2(e) / 2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.27273172829f, s, 1.0f));  // p / sqrt(log2 e) ; v_rcp_f32, 1 ulp
    e = __builtin_amdgcn_exp2f(-s * s);
    float q = fmaf(0.5307027145f, t, -0.7265760135f);                      // 0.5 * A&S coefficients
    q = fmaf(q, t, 0.7107068705f);
    q = fmaf(q, t, -0.142248368f);
    q = fmaf(q, t, 0.127414796f);
    qt = q * t;
}
__device__ __forceinline__ float gelu_f(float x) {
    float ax, e, qt;
    gelu_core(x, ax, e, qt);
    return fmaf(ax, fmaf(-qt, e, 0.5f), 0.5f * x);       // max(x,0) = 0.5 (x + |x|): no NaN-canonicalising v_max
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float ax, e, qt;
    gelu_core(x, ax, e, qt);
    const float w = e * fmaf(ax, 0.3989422804014327f, -qt);
    return 0.5f + copysignf(0.5f + w, x);
}
// both at once (the backward epilogues need GELU(h) for the weight gradient and GELU'(h) for the data gradient)
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
    float ax, e, qt;
    gelu_core(x, ax, e, qt);
    g = fmaf(ax, fmaf(-qt, e, 0.5f), 0.5f * x);
    const float w = e * fmaf(ax, 0.3989422804014327f, -qt);
    dg = 0.5f + copysignf(0.5f + w, x);
}
// (kept for callers that want the two factors) cdf = Phi(x), pdf = phi(x)
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    float ax, e, qt;
    gelu_core(x, ax, e, qt);
    const float tail = qt * e;
    cdf = x >= 0.f ? 1.0f - tail : tail;
    pdf = 0.3989422804014327f * e;
}

// erf-GELU where the result is ROUNDED TO BF16 (fused CNBlock MLP, bf16 GEMM epilogues): a polynomial with no transcendental
// for the activation and one for its derivative.  Measured issue cost on gfx950 (tools/micro/valu_rate2.hip, >= 2 waves per SIMD):
// v_exp_f32 / v_rcp_f32 = 3.4 x a v_fma_f32, so gelu_f above costs 11 + 2 * 3.4 = 17.8 fma-slots and this form 11; gelu_both
// 22.8 against 18.4.  (Packed forms do not help: v_pk_fma_f32 = 3.9 slots per 2 elements, v_pk_fma_f16 = 1.65.)
//     xc = clamp(x, -4, 4),  u = xc^2,  t = 0.5 + xc R(u) ~ Phi(x)  (R(16) * 4 = 0.5 exactly: t = 0 / 1 beyond the clamp)
//     GELU(x) = x t,   GELU'(x) = t + xc phi(xc),  phi(xc) = 2^(-u log2(e) / 2) / sqrt(2 pi)
// R: degree 6 in u (degree 7 until round 3), fitted by tools/fit_gelu.py.  Error of the fp32 evaluation over [-12, 12] against the exact
// function: relative <= 5.8e-4 for x > 0, absolute <= 2.4e-4 for x < 0 (where |GELU| <= 0.17) - a bf16 result has a relative
// rounding error of up to 2^-8 = 3.9e-3; GELU' absolute <= 5.4e-4.  fp32 outputs keep gelu_f / gelu_grad_f (1.5e-7).
__device__ __forceinline__ float gelu_bf16_t(float x, float& xc, float& u) {
    xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    u = xc * xc;
    // round 4: degree 6 in u (tools/fit_gelu.py, DEG = 6): one fma less per element in kernels that are bound by exactly these instructions;
    // relative <= 5.8e-4 for x > 0, absolute <= 2.4e-4 for x < 0 (degree 7: 4.3e-4 / 1.7e-4) - a quarter of the bf16 rounding of the result
    float r = fmaf(2.2038399460e-08f, u, -1.5565415077e-06f);
    r = fmaf(r, u, 4.7013063952e-05f);
    r = fmaf(r, u, -8.0343697344e-04f);
    r = fmaf(r, u, 8.7106341480e-03f);
    r = fmaf(r, u, -6.4399300920e-02f);
    r = fmaf(r, u, 3.9770520692e-01f);      // (nudged so that the fp32 chain gives R(16) = 0.125 exactly)
    return fmaf(xc, r, 0.5f);
}
__device__ __forceinline__ float gelu_bf16(float x) {
    float xc, u;
    return x * gelu_bf16_t(x, xc, u);
}
__device__ __forceinline__ void gelu_bf16_both(float x, float& g, float& dg) {
    float xc, u;
    const float t = gelu_bf16_t(x, xc, u);
    g = x * t;
    const float e = __builtin_amdgcn_exp2f(u * -0.72134752044448170f);      // exp(-xc^2 / 2)
    dg = fmaf(xc * e, 0.3989422804014327f, t);
}
// GELU'(x) ~ 0.5 + xc S(u), u = xc^2, xc = clamp(x, -4, 4): the derivative as ONE polynomial, no exp (round 3: the GELU' block of the
// on-chip weight-gradient backward is VALU-issue bound).  S of degree 7 in u, Lawson-reweighted minimax fit of (GELU'(x) - 0.5) / x
// with S(16) * 4 = 0.5 exactly in fp32 (derivative exactly 0 / 1 beyond the clamp); fp32 Horner over [-12, 12]: |error| <= 5.2e-4,
// all of it the clamp itself (GELU'(4) = 1.0005) - the same bound gelu_bf16_both reaches with a polynomial AND an exp.
__device__ __forceinline__ float gelu_bf16_grad_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    const float u = xc * xc;
    float r = fmaf(-2.1511327673e-08f, u, 1.5282923286e-06f);
    r = fmaf(r, u, -4.6282682125e-05f);
    r = fmaf(r, u, 7.8770984093e-04f);
    r = fmaf(r, u, -8.3827432669e-03f);
    r = fmaf(r, u, 5.8451897744e-02f);
    r = fmaf(r, u, -2.6627910133e-01f);
    r = fmaf(r, u, 7.9896733648e-01f);
    return fmaf(xc, r, 0.5f);
}
__device__ __forceinline__ float gelu_bf16_grad(float x) {
    float g, dg;
    gelu_bf16_both(x, g, dg);
    return dg;
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
