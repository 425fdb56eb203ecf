// Issue-rate probe: v_dot2c_f32_bf16 vs v_fma_f32 vs v_perm_b32 (wave64, 4 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; u[i] = 0x3f803f80u + i + threadIdx.x; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
            if (MODE == 1) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, u[i]), __builtin_bit_cast(bf2, u[(i + 1) & 7]), a[i], false);
            if (MODE == 2) u[i] = __builtin_amdgcn_perm(u[i], u[(i + 3) & 7], 0x05040100u);
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
    float* out; (void)hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4096, grid = 256 * 4;
    k<MODE><<<grid, 256>>>(out, 16);
    (void)hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-8s %.3f ms\n", name, ms);
}
int main() { run<0>("fma"); run<1>("dot2c"); run<2>("perm"); return 0; }
