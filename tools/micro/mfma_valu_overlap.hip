// Does the matrix pipe overlap with plain VALU work on gfx950, within one wave and across waves of a SIMD?
// Loop bodies of 4 MFMAs (16x16x32 bf16, 4 independent accumulators) and/or V fp32 FMAs (8 independent chains), in explicit
// program order (asm volatile is not reordered).  Prints shader cycles (s_memtime) per loop body as seen by one wave, and the
// wall-clock figure per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))

// MODE 0: MFMA only   1: VALU only (16/body)   2: 1 MFMA + 4 VALU interleaved   3: 4 MFMA then 16 VALU
// MODE 4: 1 MFMA + 8 VALU interleaved (32/body)   5: VALU only (32/body)
// MODE 6: waves 0..3 (8..11) of the workgroup run MODE 0, waves 4..7 (12..15) run MODE 1 (wave specialisation)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 7); b[i] = (short)(0x3c00 + i); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float c1 = 0.999f, c2 = 0.001f;
    const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4 || (MODE == 6 && ((threadIdx.x >> 8) & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 5 || (MODE == 6 && ((threadIdx.x >> 8) & 1) == 1);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 6 && do_mfma)) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
        if (MODE == 1 || (MODE == 6 && do_valu)) { for (int r = 0; r < 2; ++r) { FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]); } }
        if (MODE == 5) { for (int r = 0; r < 4; ++r) { FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]); } }
        if (MODE == 2) {
            MFMA(acc0); FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]);
            MFMA(acc1); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]);
            MFMA(acc2); FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]);
            MFMA(acc3); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]);
        }
        if (MODE == 3) {
            MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3);
            for (int r = 0; r < 2; ++r) { FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]); }
        }
        if (MODE == 4) {
            MFMA(acc0); FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]);
            MFMA(acc1); FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]);
            MFMA(acc2); FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]);
            MFMA(acc3); FMA(v[0]); FMA(v[1]); FMA(v[2]); FMA(v[3]); FMA(v[4]); FMA(v[5]); FMA(v[6]); FMA(v[7]);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        cyc[threadIdx.x >> 6] = ((t1 - t0) << 4) | ((hwid >> 4) & 3);      // SIMD_ID = bits 5:4
    }
}

template <int MODE> void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, (size_t)1024 * 256 * 4); hipMalloc(&cyc, 128);
    hipMemset(cyc, 0, 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 8192, grid = 256;                       // one workgroup per CU: threads / 256 waves per SIMD
    k<MODE><<<grid, threads>>>(out, cyc, 64);
    hipEventRecord(e0); k<MODE><<<grid, threads>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    printf("%-22s %d waves/SIMD %7.3f ms | cycles per body (simd):", name, threads / 256, ms);
    for (int w = 0; w < nw; ++w) printf(" %.0f(%d)", (double)(h[w] >> 4) / iters, (int)(h[w] & 15));
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int t : {256, 512, 768, 1024}) {
        run<0>("4 MFMA", t); run<1>("16 FMA", t); run<5>("32 FMA", t); run<2>("4 x (MFMA + 4 FMA)", t); run<3>("4 MFMA ; 16 FMA", t); run<4>("4 x (MFMA + 8 FMA)", t);
    }
    run<6>("MFMA waves | FMA waves", 512);
    run<6>("MFMA waves | FMA waves", 1024);
    return 0;
}
