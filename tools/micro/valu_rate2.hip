// Issue-rate probe, round 2: packed f16 / mixed-precision ops considered for a cheaper GELU (wave64; 2 and 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8];
    h2 h[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; h[i] = h2{(_Float16)(a[i] * 0.01f), (_Float16)(a[i] * 0.02f)}; }
    const h2 c1 = {(_Float16)0.999f, (_Float16)0.998f}, c2 = {(_Float16)0.001f, (_Float16)0.002f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
            if (MODE == 1) h[i] = h[i] * c1 + c2;                                        // v_pk_fma_f16
            if (MODE == 2) a[i] = __builtin_fmaf(a[i], (float)h[i].x, 0.001f);           // v_fma_mix_f32 (f16 source)
            if (MODE == 3) { h[i] = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a[i], a[(i + 1) & 7])); a[i] += 0.5f; }   // cvt_pkrtz + add
            if (MODE == 4) a[i] = __builtin_fmaxf(a[i] * 0.999f, 0.f);                   // mul + max
            if (MODE == 5) a[i] = __builtin_amdgcn_fmed3f(a[i], 0.1f, 9.f) * 0.999f;     // med3 + mul
            if (MODE == 6) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[i], a[(i + 1) & 7]}; v = v * f2{0.999f, 0.998f} + f2{0.001f, 0.002f}; a[i] = v.x; a[(i + 1) & 7] = v.y; }  // v_pk_fma_f32
            if (MODE == 7) a[i] = __builtin_amdgcn_exp2f(a[i]);
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)h[i].x + (float)h[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int wgs_per_cu) {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, grid = 256 * wgs_per_cu;          // 256-thread WGs: wgs_per_cu waves per SIMD
    k<MODE><<<grid, 256>>>(out, 16);
    hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double groups = (double)grid * 4 * iters * 8;             // loop bodies over all waves
    double clk = ms * 1e-3 * 2.4e9;
    printf("%-14s %d waves/SIMD  %.3f ms -> %.2f clk (at 2.4 GHz) per loop body per SIMD\n", name, wgs_per_cu, ms, clk / (groups / 1024));
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("fma_f32", 1); run<1>("pk_fma_f16", 1); run<2>("fma_mix", 1); run<3>("cvt_pkrtz+add", 1); run<4>("mul+max", 1); run<5>("med3+mul", 1); run<6>("pk_fma_f32", 1); run<7>("exp2", 1); }
        if (w == 2) { run<0>("fma_f32", 2); run<1>("pk_fma_f16", 2); run<2>("fma_mix", 2); run<3>("cvt_pkrtz+add", 2); run<4>("mul+max", 2); run<5>("med3+mul", 2); run<6>("pk_fma_f32", 2); run<7>("exp2", 2); }
        if (w == 4) { run<0>("fma_f32", 4); run<1>("pk_fma_f16", 4); run<2>("fma_mix", 4); run<3>("cvt_pkrtz+add", 4); run<4>("mul+max", 4); run<5>("med3+mul", 4); run<6>("pk_fma_f32", 4); run<7>("exp2", 4); }
    }
    return 0;
}
