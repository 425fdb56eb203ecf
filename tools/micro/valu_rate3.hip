// Issue cost of plain VALU instructions by operand kind on gfx950 (4 waves per SIMD, 16 independent instructions per loop body).
// Found with mfma_valu_coexec: v_fma_f32 with three VGPR sources issues every ~2.7 cycles, the same instruction with an SGPR source
// every ~4.4.  This probe classifies the instructions the kernels use.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate3 valu_rate3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int V> __device__ __forceinline__ void op(float& x, float& y, f32x2& p, f32x2& q, float a, float b, float cs, f32x2 sp, unsigned long long& msk) {
    if constexpr (V == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    if constexpr (V == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    if constexpr (V == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    if constexpr (V == 4) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x) : "v"(a));
    if constexpr (V == 5) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(x) : "v"(a));
    if constexpr (V == 6) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(x) : "v"(b));
    if constexpr (V == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));
    if constexpr (V == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    if constexpr (V == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a));
    if constexpr (V == 11) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x));
    if constexpr (V == 12) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x));
    if constexpr (V == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if constexpr (V == 14) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x));
    if constexpr (V == 15) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(cs));
    if constexpr (V == 16) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    if constexpr (V == 17) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    if constexpr (V == 18) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 19) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if constexpr (V == 20) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 21) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    if constexpr (V == 22) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
    if constexpr (V == 23) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 24) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(a));
    if constexpr (V == 25) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    if constexpr (V == 26) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x));
    if constexpr (V == 27) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 28) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    if constexpr (V == 29) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 30) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if constexpr (V == 32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p) : "s"(sp), "v"(q));
    if constexpr (V == 33) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(p) : "v"(q), "s"(sp));
    if constexpr (V == 34) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p) : "s"(sp));
    if constexpr (V == 35) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(cs));
    if constexpr (V == 36) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p) : "v"(q), "v"(q));
    if constexpr (V == 37) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(msk));
    if constexpr (V == 38) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
    if constexpr (V == 39) asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(x), "v"(a) : "vcc");
    if constexpr (V == 40) asm volatile("v_cmp_gt_f32_e64 %2, %1, %0\n v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(msk));
    if constexpr (V == 31) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a));
    // round 3: two bf16 multiply-adds per lane and instruction (depthwise 7x7 on pixel pairs?)
    if constexpr (V == 41) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 42) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(y), "v"(b));
    if constexpr (V == 43) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (V == 44) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x) : "v"(a));
}

template <int V>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    float v[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = 1.0f + threadIdx.x * 1e-3f + i; p[i] = f32x2{v[i], v[i] + 0.5f}; }
    float a = 0.999f + threadIdx.x * 1e-9f, b = 0.001f + threadIdx.x * 1e-9f;
    asm volatile("" : "+v"(a), "+v"(b));
    const float cs = __builtin_amdgcn_readfirstlane(0.999f);
    f32x2 sp = {cs, cs};
    asm volatile("" : "+s"(sp));
    unsigned long long msk = __builtin_amdgcn_read_exec() ^ 0x5555;
    asm volatile("" : "+s"(msk));
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) op<V>(v[i], v[(i + 3) & 7], p[i], p[(i + 3) & 7], a, b, cs, sp, msk);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int V> void row(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, (size_t)1024 * 256 * 4); (void)hipMalloc(&cyc, 128);
    (void)hipMemset(cyc, 0, 128);
    const int iters = 4096;
    k<V><<<256, 1024>>>(out, cyc, 64);
    k<V><<<256, 1024>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[16]; (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    long long mx = 0; for (int w = 0; w < 16; ++w) mx = h[w] > mx ? h[w] : mx;
    (void)hipFree(out); (void)hipFree(cyc);
    printf("%-36s %5.2f cycles per instruction per SIMD\n", name, (double)mx / iters / 4 / 16);
}
int main() {
    row<0>("v_fma_f32 x,x,v,v"); row<27>("v_fma_f32 x,v,v,x"); row<25>("v_fma_f32 x,x,x,x"); row<29>("v_fma_f32 x,|x|,v,v"); row<4>("v_fma_f32 x,x,v,0.5 (inline)");
    row<5>("v_fmaak_f32 (literal)"); row<6>("v_fmamk_f32 (literal)"); row<7>("v_fmac_f32 x,v,v");
    row<1>("v_mul_f32 x,x,v"); row<26>("v_mul_f32 x,x,x"); row<14>("v_mul_f32 x,0.5,x (inline)"); row<15>("v_mul_f32 x,s,x");
    row<2>("v_add_f32 x,x,v"); row<16>("v_sub_f32 x,x,v"); row<3>("v_max_f32 x,x,v"); row<17>("v_min_f32 x,x,v"); row<18>("v_med3_f32 x,x,v,v");
    row<8>("v_pk_fma_f32"); row<36>("v_pk_fma_f32 op_sel_hi (v only)"); row<32>("v_pk_fma_f32 p,p,s,q"); row<33>("v_pk_fma_f32 p,p,q,s"); row<34>("v_pk_mul_f32 p,p,s"); row<35>("v_fma_f32 x,x,v,s"); row<9>("v_pk_mul_f32"); row<28>("v_pk_add_f32"); row<23>("v_pk_fma_f16"); row<24>("v_pk_mul_f16");
    row<10>("v_and_b32 x,x,v"); row<12>("v_and_b32 x,literal,x"); row<11>("v_lshlrev_b32 x,16,x"); row<20>("v_perm_b32"); row<21>("v_add_u32 x,x,v");
    row<22>("v_mov_b32"); row<31>("v_cndmask_b32 x,x,v,vcc"); row<37>("v_cndmask_b32_e64 x,x,v,s[pair]"); row<39>("v_cmp_gt_f32 vcc"); row<38>("v_cmp vcc + v_cndmask vcc (pair)"); row<40>("v_cmp_e64 s + v_cndmask_e64 s (pair)"); row<13>("v_cvt_pk_bf16_f32"); row<19>("v_rcp_f32"); row<30>("v_exp_f32");
    row<41>("v_dot2c_f32_bf16 x,v,v"); row<42>("v_dot2c_f32_bf16 x,y,v"); row<43>("v_dot2_f32_bf16 x,v,v,x"); row<44>("v_alignbit_b32 x,x,v,16");
    return 0;
}
