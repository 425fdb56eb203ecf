// Issue-rate probe for the ops inside the GELU epilogue: fma vs v_exp_f32 vs v_rcp_f32 (wave64, 4 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
            if (MODE == 1) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.f + a[i];      // exp + fma
            if (MODE == 2) a[i] = __builtin_amdgcn_rcpf(a[i]) * 0.f + a[i];       // rcp + fma
            if (MODE == 3) { a[i] = __builtin_amdgcn_exp2f(a[i]); }
            if (MODE == 4) { a[i] = __builtin_amdgcn_rcpf(a[i]); }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int ops_per) {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, grid = 256 * 4;   // 4 WGs per CU = 4 waves per SIMD
    k<MODE><<<grid, 256>>>(out, 16);
    hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)grid * 4 /*waves*/ * iters * 8;       // wave-level loop bodies
    double clk = ms * 1e-3 * 2.4e9;
    printf("%-10s %.3f ms  -> %.2f clk per wave-instr-group per SIMD (%d instr each)\n", name, ms, clk / (inst / 1024), ops_per);
}
int main() { run<0>("fma", 1); run<1>("exp+fma", 2); run<2>("rcp+fma", 2); run<3>("exp", 1); run<4>("rcp", 1); return 0; }
