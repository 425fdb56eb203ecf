// Which plain-VALU instructions co-execute with the matrix pipe on gfx950?  (SQ_VALU_MFMA_COEXEC_CYCLES exists, so some do.)
// 4 waves per SIMD (1024-thread workgroups, one per CU).  Per loop body: 4 MFMAs and/or 16 VALU instructions of one kind, in
// explicit program order.  Printed: cycles per body per SIMD for MFMA only, VALU only, both (interleaved 1 + 4) - "both" close to
// max(a, b) means co-execution, close to a + b means the two serialise.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_coexec mfma_valu_coexec.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int V> __device__ __forceinline__ void valu(float& x, float& y, float c1v, float c2v, float cs) {
    if constexpr (V == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1v), "v"(c2v));          // 3 VGPR sources
    if constexpr (V == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(cs));                          // 1 VGPR source
    if constexpr (V == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "s"(cs));                      // 1 distinct VGPR
    if constexpr (V == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
    if constexpr (V == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(cs));
    if constexpr (V == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if constexpr (V == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if constexpr (V == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(cs), "v"(c2v));          // 2 VGPR sources
    if constexpr (V == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1v), "v"(c2v));
    if constexpr (V == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&x)) : "v"(*reinterpret_cast<double*>(&y)));
}

// WHAT: 1 = MFMA only, 2 = VALU only, 3 = both;  BIG: 32x32x16 (8 passes) instead of 16x16x32 (4 passes)
template <int V, int WHAT, bool BIG>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 7); b[i] = (short)(0x3c00 + i); }
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    f32x16 g0 = {}, g1 = {};
    __attribute__((aligned(8))) float v[10];
    for (int i = 0; i < 10; ++i) v[i] = 1.0f + threadIdx.x * 1e-3f + i;
    const float c1v = 0.999f + threadIdx.x * 1e-9f, c2v = 0.001f;
    const float cs = __builtin_amdgcn_readfirstlane(0.999f);
#define MF4(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF8(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VA(i) valu<V>(v[i], v[(i) + 2], c1v, c2v, cs)
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (!BIG) {
            if (WHAT & 1) MF4(s0); if (WHAT & 2) { VA(0); VA(1); VA(2); VA(3); }
            if (WHAT & 1) MF4(s1); if (WHAT & 2) { VA(4); VA(5); VA(6); VA(7); }
            if (WHAT & 1) MF4(s2); if (WHAT & 2) { VA(0); VA(1); VA(2); VA(3); }
            if (WHAT & 1) MF4(s3); if (WHAT & 2) { VA(4); VA(5); VA(6); VA(7); }
        } else {            // 2 MFMAs of 8 passes = the same matrix-pipe time; 8 VALU after each
            if (WHAT & 1) MF8(g0); if (WHAT & 2) { VA(0); VA(1); VA(2); VA(3); VA(4); VA(5); VA(6); VA(7); }
            if (WHAT & 1) MF8(g1); if (WHAT & 2) { VA(0); VA(1); VA(2); VA(3); VA(4); VA(5); VA(6); VA(7); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 10; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += s0[i] + s1[i] + s2[i] + s3[i];
    for (int i = 0; i < 16; ++i) s += g0[i] + g1[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int V, int WHAT, bool BIG> double one() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, (size_t)1024 * 256 * 4); (void)hipMalloc(&cyc, 128);
    (void)hipMemset(cyc, 0, 128);
    const int iters = 4096;
    k<V, WHAT, BIG><<<256, 1024>>>(out, cyc, 64);
    k<V, WHAT, BIG><<<256, 1024>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[16]; (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    long long mx = 0; for (int w = 0; w < 16; ++w) mx = h[w] > mx ? h[w] : mx;
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)mx / iters / 4;          // 4 waves per SIMD ran `iters` bodies each
}
template <int V> void row(const char* name) {
    const double m4 = one<V, 1, false>(), va = one<V, 2, false>(), b4 = one<V, 3, false>();
    const double m8 = one<V, 1, true>(), b8 = one<V, 3, true>();
    printf("%-30s VALU x16 %6.1f | 16x16x32: mfma %5.1f both %6.1f (sum %6.1f) | 32x32x16: mfma %5.1f both %6.1f (sum %6.1f)\n", name, va, m4, b4, m4 + va, m8, b8, m8 + va);
}
int main() {
    printf("cycles per loop body per SIMD (4 waves per SIMD); body = 4 MFMA 16x16x32 (or 2 MFMA 32x32x16) and / or 16 VALU\n");
    row<0>("v_fma_f32 v,v,v,v");
    row<7>("v_fma_f32 v,v,s,v");
    row<2>("v_fma_f32 v,v,s,v(same)");
    row<1>("v_mul_f32 v,s,v");
    row<3>("v_mov_b32");
    row<4>("v_add_u32 v,s,v");
    row<5>("v_cvt_pk_bf16_f32");
    row<8>("v_med3_f32 v,v,v,v");
    row<6>("v_exp_f32");
    row<9>("v_pk_fma_f32");
    return 0;
}
