// bf16 MFMA GEMMs for the encoder towers (gfx950).
//
//   mmg_gemm_nt_bf16 : C[M,N]   = epilogue( A[M,K] * B[N,K]^T )          forward linears, data gradients
//   mmg_gemm_tn_bf16 : C[N1,N2] += A[M,N1]^T * B[M,N2]   (fp32, atomic)  weight gradients
//
// These replace the ATen/cuBLAS calls behind nn.Linear in the reference's towers:
//   HF BertLayer Q/K/V/out/FFN linears  (mmgclip/networks/encoder.py:138,156),
//   torchvision CNBlock Linear(C,4C)/Linear(4C,C) and the 2x2/4x4 patchify convolutions of ConvNeXt
//   (mmgclip/networks/encoder.py:53; module tree in notebooks/clf_convnext_tiny_experimental.ipynb cell 3),
//   nn.Linear of the projection heads (mmgclip/networks/projection.py:17,42-49,88-90).
//
// Design (CDNA4): 256 threads = 4 waves as 2x2, v_mfma_f32_16x16x32_bf16, operands staged HBM -> LDS with
// 16-byte global_load_lds (no VGPR round trip), double-buffered, one barrier per K tile.  The LDS image is
// lane-linear (a requirement of LDS-DMA), so the bank-conflict swizzle is applied to the per-lane SOURCE
// address and again on the fragment read (an XOR involution).  The MFMA is issued swapped (A := weight
// fragment, B := activation fragment) so a lane ends up with 4 CONSECUTIVE output columns of one row; the
// accumulators go through LDS once and leave as whole 16-byte row segments with bias / GELU / layer-scale /
// residual applied on the way (no separate elementwise kernels, no fp32 round trip through HBM).
// The weight-gradient kernel reads both operands with ds_read_b64_tr_b16 (hardware transpose) because the
// reduction index (the row m) is the slow index of both inputs.
#include "common.h"
#include <type_traits>
#include "gemm_tn.h"
#include <stdlib.h>

#define GEMM_THREADS 256

// EPI_DGELU_ONLY: the data gradient alone, C = (A B^T) * GELU'(aux_in) - for callers whose forward kept GELU(h) (round 3: the epilogue of
// EPI_DGELU is VALU-bound and half of its arithmetic and stores rebuild that activation)
// Round 4: EPI_GELU_DAUX = GELU forward whose side output is GELU'(pre-activation) instead of the pre-activation, and
// EPI_MUL_AUX = C = (A B^T) * aux_in - the data gradient of a layer whose forward kept GELU' (one multiply per element where EPI_DGELU_ONLY
// evaluated a degree-7 polynomial: 19 000 of that tile's 53 600 cycles were this arithmetic, tools/nt_probe.py).
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_RELU = 3, EPI_DRELU = 4, EPI_DGELU_ONLY = 5, EPI_GELU_DAUX = 6, EPI_MUL_AUX = 7 };

struct GemmNT {
    const bf16_t* A; const bf16_t* B;
    int M, N, K, lda, ldb;
    void* C; int ldc; int out_f32;
    const float* bias;        // [N]  added to the accumulator
    const float* colscale;    // [N]  multiplies after activation (ConvNeXt layer scale)
    const bf16_t* residual; int ldr;   // [M,N] added last
    const bf16_t* aux_in; int ldai;    // [M,N] pre-activation for EPI_DGELU / EPI_DGELU_ONLY / EPI_DRELU
    bf16_t* aux_out; int ldao;         // [M,N] receives the pre-activation for EPI_GELU / EPI_RELU (may be null)
    int epi;
    float alpha;
    int tiles_m, tiles_n;
    int nt_store;             // stream the outputs past L2 (they are not re-read before they would be evicted anyway)
    // fp8 path (mmg_gemm_nt_fp8): A / B hold OCP e4m3 bytes; K, lda, ldb are passed here in 2-byte units
    int out_fp8;              // C receives 8-bit floats (saturating), ldc in bytes: 1 = e4m3, 2 = e5m2
    const float* alpha_dev;   // optional device scalar multiplied into alpha (1 / weight scale, produced on the device)
    const float* alpha_dev2;  // a second one (1 / gradient scale of the fp8 backward)
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XOR applied to the 16-byte chunk index of row r (see the bank analysis in DESIGN.md §kernels/GEMM)
template <int BK>
__device__ __forceinline__ int swz(int r) {
    if (BK == 64) return r & 7;
    return (4 - ((r >> 2) & 3)) & 3;   // BK == 32
}

// Stage ROWS x BK bf16 (rows row0.. of G, clamped to rows_total-1) into a lane-linear LDS tile.
template <int ROWS, int BK, int THREADS = GEMM_THREADS>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ G, int ld, int row0, int rows_total, int k0,
                                           char* lds_tile, int tid) {
    constexpr int CPR = BK / 8;
    constexpr int NCH = ROWS * CPR;
#pragma unroll
    for (int it = 0; it < (NCH + THREADS - 1) / THREADS; ++it) {
        const int p = it * THREADS + tid;
        if (NCH % THREADS == 0 || p < NCH) {
            const int r = p / CPR, s = p % CPR;
            const int c = s ^ swz<BK>(r);
            const int gr = min(row0 + r, rows_total - 1);
            glds16(G + (size_t)gr * ld + k0 + c * 8, lds_tile + (size_t)(it * THREADS + (tid & ~63)) * 16);
        }
    }
}

template <int BK>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * (BK * 2) + ((c ^ swz<BK>(r)) << 4));
}

// XCD-aware tile order: workgroups that share an A row panel run on one XCD (same L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, l = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}

__device__ __forceinline__ void store16(void* dst, const uint4 v, int nt) {
    if (nt) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(dst));
    } else {
        *reinterpret_cast<uint4*>(dst) = v;
    }
}

// One output row segment of 8 columns: staged fp32 accumulators -> bias / activation / layer scale / residual -> store.
// EPI is a template parameter (round 3): with the runtime `g.epi` tested inside the element loops, hipcc kept one scalar branch per element
// PAIR - 64 basic blocks per thread and tile, each a single dependent v_exp -> v_rcp -> fma chain with nothing to interleave (tools/nt_probe.py:
// 26 000 of the 67 700 cycles of a 256 x 256 x 384 GELU' tile were this arithmetic).  The kernel now switches once per tile (nt_epilogue_slabs)
// and the eight columns of a row are straight-line code.
// RES: 1 = a residual row is added, 0 = none, -1 = g.residual decides at run time.
template <int EPI, int RES = -1>
__device__ __forceinline__ void nt_epilogue_row(const GemmNT& g, const float* crow, int gr, int gc, const float* bias,
                                                const float* cs, const uint4 res, const uint4 aux, float alpha) {
    float v[8];
    const f32x4 lo = *reinterpret_cast<const f32x4*>(crow);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(crow + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], alpha, bias[e]);
    if constexpr (EPI == EPI_GELU_DAUX) {
        // GELU(v) out, GELU'(v) to aux_out (bf16 outputs: the two exp-free polynomials of common.h; fp32 / exact builds: the shared rcp / exp form)
        float dv[8];
#ifndef NT_GELU_EXACT
        if (!g.out_f32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { dv[e] = gelu_bf16_grad_poly(v[e]); v[e] = gelu_bf16(v[e]); }
        } else
#endif
        {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float a, d; gelu_both(v[e], a, d); v[e] = a; dv[e] = d; }
        }
        if (g.aux_out)
            store16(g.aux_out + (size_t)gr * g.ldao + gc, make_uint4(pack2bf(dv[0], dv[1]), pack2bf(dv[2], dv[3]), pack2bf(dv[4], dv[5]), pack2bf(dv[6], dv[7])), g.nt_store);
    } else if constexpr (EPI == EPI_MUL_AUX) {
        const unsigned hw[4] = {aux.x, aux.y, aux.z, aux.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] *= bf2f_lo(hw[e]); v[2 * e + 1] *= bf2f_hi(hw[e]); }
    } else if constexpr (EPI == EPI_GELU || EPI == EPI_RELU) {
        if (g.aux_out) {
            uint4 o;
            o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
            store16(g.aux_out + (size_t)gr * g.ldao + gc, o, g.nt_store);
        }
#ifndef NT_GELU_EXACT        // bf16 / fp8 outputs: the exp-free polynomial of common.h (2^-11 relative, the output rounds at 2^-9); fp32 outputs stay exact
        if (EPI == EPI_GELU && !g.out_f32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_bf16(v[e]);
        } else
#endif
        {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (EPI == EPI_GELU) ? gelu_f(v[e]) : fmaxf(v[e], 0.f);
        }
    } else if constexpr (EPI == EPI_DGELU || EPI == EPI_DGELU_ONLY || EPI == EPI_DRELU) {
        const unsigned hw[4] = {aux.x, aux.y, aux.z, aux.w};
        unsigned act[4];
        // the activation itself (needed by the weight-gradient GEMM of the same layer) is emitted beside its derivative: that replaces a separate
        // read-modify-write pass over the [M,N] tensor.  bf16 / fp8 data gradients: two exp-free polynomials (common.h - the pair the fused CNBlock
        // kernels use; 2^-11 against an output that rounds at 2^-9); fp32 data gradients: the shared rcp / exp form (1.5e-7).  -DNT_GELU_EXACT: always.
        auto pairs = [&](auto poly_tag) {
            constexpr bool POLY = decltype(poly_tag)::value;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h0 = bf2f_lo(hw[e]), h1 = bf2f_hi(hw[e]);
                if constexpr (EPI == EPI_DGELU_ONLY) {
                    v[2 * e] *= POLY ? gelu_bf16_grad_poly(h0) : gelu_grad_f(h0);
                    v[2 * e + 1] *= POLY ? gelu_bf16_grad_poly(h1) : gelu_grad_f(h1);
                } else if constexpr (EPI == EPI_DGELU) {
                    float a0, d0, a1, d1;
                    if constexpr (POLY) {
                        a0 = gelu_bf16(h0); d0 = gelu_bf16_grad_poly(h0);
                        a1 = gelu_bf16(h1); d1 = gelu_bf16_grad_poly(h1);
                    } else {
                        gelu_both(h0, a0, d0);
                        gelu_both(h1, a1, d1);
                    }
                    v[2 * e] *= d0;
                    v[2 * e + 1] *= d1;
                    act[e] = pack2bf(a0, a1);
                } else {
                    act[e] = pack2bf(fmaxf(h0, 0.f), fmaxf(h1, 0.f));
                    v[2 * e] = h0 > 0.f ? v[2 * e] : 0.f;
                    v[2 * e + 1] = h1 > 0.f ? v[2 * e + 1] : 0.f;
                }
            }
        };
#ifndef NT_GELU_EXACT
        if ((EPI == EPI_DGELU || EPI == EPI_DGELU_ONLY) && !g.out_f32) pairs(std::true_type{}); else
#endif
        pairs(std::false_type{});
        if constexpr (EPI != EPI_DGELU_ONLY)
            if (g.aux_out) store16(g.aux_out + (size_t)gr * g.ldao + gc, make_uint4(act[0], act[1], act[2], act[3]), g.nt_store);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= cs[e];
    if (RES == 1 || (RES < 0 && g.residual)) {
        const unsigned rw[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += bf2f_lo(rw[e]); v[2 * e + 1] += bf2f_hi(rw[e]); }
    }
    if (g.out_fp8 == 2) {
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(g.C) + (size_t)gr * g.ldc + gc) =
            make_uint2(pack4_e5m2(v[0], v[1], v[2], v[3]), pack4_e5m2(v[4], v[5], v[6], v[7]));
    } else if (g.out_fp8) {
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(g.C) + (size_t)gr * g.ldc + gc) =
            make_uint2(pack4_e4m3(v[0], v[1], v[2], v[3]), pack4_e4m3(v[4], v[5], v[6], v[7]));
    } else if (g.out_f32) {
        float* dst = reinterpret_cast<float*>(g.C) + (size_t)gr * g.ldc + gc;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        store16(reinterpret_cast<bf16_t*>(g.C) + (size_t)gr * g.ldc + gc, o, g.nt_store);
    }
}
// the same with the mode read at run time (kernels whose epilogue is not on a hot path)
__device__ __forceinline__ void nt_epilogue_row_rt(const GemmNT& g, const float* crow, int gr, int gc, const float* bias,
                                                   const float* cs, const uint4 res, const uint4 aux, float alpha) {
    switch (g.epi) {
        case EPI_GELU: nt_epilogue_row<EPI_GELU>(g, crow, gr, gc, bias, cs, res, aux, alpha); break;
        case EPI_DGELU: nt_epilogue_row<EPI_DGELU>(g, crow, gr, gc, bias, cs, res, aux, alpha); break;
        case EPI_RELU: nt_epilogue_row<EPI_RELU>(g, crow, gr, gc, bias, cs, res, aux, alpha); break;
        case EPI_DRELU: nt_epilogue_row<EPI_DRELU>(g, crow, gr, gc, bias, cs, res, aux, alpha); break;
        case EPI_DGELU_ONLY: nt_epilogue_row<EPI_DGELU_ONLY>(g, crow, gr, gc, bias, cs, res, aux, alpha); break;
        case EPI_GELU_DAUX: nt_epilogue_row<EPI_GELU_DAUX>(g, crow, gr, gc, bias, cs, res, aux, alpha); break;
        case EPI_MUL_AUX: nt_epilogue_row<EPI_MUL_AUX>(g, crow, gr, gc, bias, cs, res, aux, alpha); break;
        default: nt_epilogue_row<EPI_NONE>(g, crow, gr, gc, bias, cs, res, aux, alpha); break;
    }
}

// ---------------------------------------------------------------------------------------------
// NT kernel.  WAVES_M x 2 waves, wave tile 64 x BN/2, BM = 64 * WAVES_M, NST staging buffers.
//   <128, BN, BK, 2, 2> : 256 threads, two workgroups per CU (small / skinny problems, K % 64 != 0)
//   <256, 128, 64, 4, 3>: 512 threads, one workgroup per CU, two K tiles in flight across the barrier
//                         (counted s_waitcnt vmcnt + raw s_barrier; a __syncthreads() would drain the LDS-DMA queue)
// The main loop is unrolled over the NST buffers so that every LDS address is "per-thread constant + immediate" and
// every global source address is "uniform 64-bit base (SGPR, advanced per K tile) + per-thread 32-bit offset":
// no vector ALU work per K tile besides the MFMAs (measured before: ~90 non-MFMA instructions per tile and wave).
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef NT_PROBE
// Debug builds only (tools/nt_probe.py): shader-clock totals per phase of gemm_nt_kernel, summed over waves
//   0 whole kernel  1 set-up (offsets, per-column vectors, first stages issued)  2 main loop  3 epilogue: barriers + accumulators -> LDS
//   5 epilogue: row arithmetic + stores issued (with the waits for the slab's global reads)  6 waves
// (one plain store per workgroup slot: 200 000 waves x 8 same-address atomics took 16 ms by themselves)
#define NT_PROBE_SLOTS 32768
__device__ unsigned long long g_nt_probe[NT_PROBE_SLOTS][8];
MMG_API int mmg_debug_nt_probe(unsigned long long* out8, int reset) {
    static unsigned long long host[NT_PROBE_SLOTS][8];
    if (out8) {
        if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_nt_probe), sizeof(host)) != hipSuccess) return 1;
        for (int i = 0; i < 8; ++i) out8[i] = 0;
        for (int s = 0; s < NT_PROBE_SLOTS; ++s) for (int i = 0; i < 8; ++i) out8[i] += host[s][i];
    }
    if (reset) {
        for (int s = 0; s < NT_PROBE_SLOTS; ++s) for (int i = 0; i < 8; ++i) host[s][i] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_nt_probe), host, sizeof(host)) != hipSuccess) return 1;
    }
    return 0;
}
#define NTP_T(var) const long long var = __builtin_readcyclecounter()
#define NTP_ADD(idx, a, b) ntp[idx] += (b) - (a)
#else
#define NTP_T(var)
#define NTP_ADD(idx, a, b)
#endif

template <int BM, int BN, int BK, int WAVES_M, int NST, int F8 = 0>
__global__ __launch_bounds__(WAVES_M * 128, BN > 128 ? 1 : ((NST == 3 && BK == 32) ? 3 : 2)) void gemm_nt_kernel(const GemmNT g) {
    static_assert(!F8 || BK == 64, "fp8 operands: one 128-byte K tile = one 16x16x128 MFMA step");      // F8: 1 = A e4m3, 2 = A e5m2 (gradients); B e4m3
    constexpr int THREADS = WAVES_M * 128;
    constexpr int MI = 4, NI = BN / 32;                  // 16x16 fragments per wave (wave tile 64 x BN/2)
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int CPR = BK / 8;                           // 16-byte chunks per staged row
    constexpr int A_IT = (BM * CPR + THREADS - 1) / THREADS, B_IT = (BN * CPR + THREADS - 1) / THREADS;
    constexpr bool B_RAGGED = (BN * CPR) % THREADS != 0;  // last B pass only on the first waves (BN = 96, BK = 32)
    constexpr int LDCS = BN + 4;                          // fp32 staging pitch
    static_assert(BM == 64 * WAVES_M, "wave tile is 64 rows");
    static_assert(!(NST > 2 && B_RAGGED), "counted vmcnt needs the same number of loads in every wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
#ifdef NT_PROBE
    long long ntp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    NTP_T(nt_t0);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue geometry: each thread owns 8 consecutive columns of a row; per-column vectors are fetched now
    constexpr int TPR = BN / 8;                           // threads per row
    constexpr int RPP = THREADS / TPR;                    // rows per pass
    constexpr int QP = (64 + RPP - 1) / RPP;              // passes per 64-row slab (= one wave row)
    constexpr int PASSES = WAVES_M * QP;
    const int tr = tid / TPR, tc = (tid % TPR) * 8;
    const int gc = n0 + tc;
    const bool col_ok = (tr < RPP) && (gc < g.N);
    float bias[8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bias[e] = (col_ok && g.bias) ? g.bias[gc + e] : 0.f;
        cs[e] = (col_ok && g.colscale) ? g.colscale[gc + e] : 1.f;
    }

    // ---- per-thread constants of the staging (global byte offsets inside the tile, rows clamped) -----------------
    unsigned a_off[A_IT], b_off[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int p = it * THREADS + tid, r = p / CPR, sl = p % CPR;
        const int gr = min(m0 + r, g.M - 1) - m0;
        a_off[it] = (unsigned)(gr * g.lda + (sl ^ swz<BK>(r)) * 8) * 2u;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int p = it * THREADS + tid, r = p / CPR, sl = p % CPR;
        const int gr = min(n0 + min(r, BN - 1), g.N - 1) - n0;
        b_off[it] = (unsigned)(gr * g.ldb + (sl ^ swz<BK>(r)) * 8) * 2u;
    }
    const bool b_last = !B_RAGGED || ((B_IT - 1) * THREADS + tid < BN * CPR);
    const char* a_base = reinterpret_cast<const char*>(g.A + (size_t)m0 * g.lda);   // uniform; advanced by BK per tile
    const char* b_base = reinterpret_cast<const char*>(g.B + (size_t)n0 * g.ldb);
    char* lds_wave = smem + (tid & ~63) * 16;                                       // wave-uniform LDS-DMA base

    auto stage = [&](int buf, int kt) {
        const char* ab = a_base + (size_t)kt * (BK * 2);
        const char* bb = b_base + (size_t)kt * (BK * 2);
        char* la = lds_wave + buf * STAGE;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) glds16(ab + a_off[it], la + it * THREADS * 16);
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            if (it + 1 < B_IT || b_last) glds16(bb + b_off[it], la + A_BYTES + it * THREADS * 16);
    };

    // ---- per-thread constants of the fragment reads: row byte offset + swizzled chunk for ks = 0 and 1 ----------
    int a_frag[BK / 32], b_frag[BK / 32];
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
        const int ra = wm * 64 + li, rb = wn * (BN / 2) + li;      // rows of fragment 0; fragment i adds 16 rows
        a_frag[ks] = ra * (BK * 2) + (((4 * ks + lg) ^ swz<BK>(ra)) << 4);
        b_frag[ks] = A_BYTES + rb * (BK * 2) + (((4 * ks + lg) ^ swz<BK>(rb)) << 4);
    }
    // (rows r and r + 16 share swz(r) for BK = 64 [r & 7]; for BK = 32 swz depends on (r >> 2) & 3, unchanged by +16)

    const int nk = g.K / BK;
    constexpr int LOADS = A_IT + B_IT;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s);

    NTP_T(nt_t1);
    for (int kt0 = 0; kt0 < nk; kt0 += NST) {
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            const int kt = kt0 + s;
            if (kt < nk) {
                // tile kt has landed once at most the NST-2 younger tiles are still outstanding (in-order retirement)
                if (NST > 2 && kt + (NST - 2) < nk) wait_vmcnt<(NST > 2 ? (NST - 2) * LOADS : 0)>();
                else wait_vmcnt<0>();
                if (NST > 2) __builtin_amdgcn_s_barrier(); else __syncthreads();
                if (kt + NST - 1 < nk) stage((s + NST - 1) % NST, kt + NST - 1);
                const char* buf = smem + s * STAGE;
                if constexpr (F8 != 0) {
                    // e4m3 operands: the 128 staged bytes of a row are ONE k-step of v_mfma_f32_16x16x128_f8f6f4 (32 bytes
                    // per lane).  A lane takes the same two 16-byte pieces the bf16 loop reads for ks = 0 and 1; A and B
                    // permute k identically, so the product is unchanged and staging / swizzle / reads stay as they are.
                    constexpr int NJ = NI > 4 ? 4 : NI;        // B fragments live at a time (8 VGPRs each)
                    i32x8 af[MI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const i32x4 lo = *reinterpret_cast<const i32x4*>(buf + a_frag[0] + i * 16 * BK * 2);
                        const i32x4 hi = *reinterpret_cast<const i32x4*>(buf + a_frag[BK / 32 - 1] + i * 16 * BK * 2);
                        af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
#pragma unroll
                    for (int j0 = 0; j0 < NI; j0 += NJ) {
                        i32x8 bfr[NJ];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const i32x4 lo = *reinterpret_cast<const i32x4*>(buf + b_frag[0] + (j0 + j) * 16 * BK * 2);
                            const i32x4 hi = *reinterpret_cast<const i32x4*>(buf + b_frag[BK / 32 - 1] + (j0 + j) * 16 * BK * 2);
                            bfr[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        }
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
                                acc[i][j0 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j0 + j], 0, F8 == 2 ? 1 : 0, 0, 0, 0, 0);   // (B = weights e4m3; A e4m3 / e5m2)
                    }
                } else {
#pragma unroll
                for (int ks = 0; ks < BK / 32; ++ks) {
                    bf16x8 af[MI], bfr[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(buf + a_frag[ks] + i * 16 * BK * 2);
#pragma unroll
                    for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(buf + b_frag[ks] + j * 16 * BK * 2);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
                }
            }
        }
    }

    NTP_T(nt_t2);
    NTP_ADD(1, nt_t0, nt_t1); NTP_ADD(2, nt_t1, nt_t2);
    // issue EVERY global read of the epilogue (residual / saved pre-activation rows of all passes) before the
    // accumulators go through LDS, so their latency overlaps the staging instead of being paid pass by pass.
    constexpr bool PREFETCH = PASSES <= 8;                // (wider tiles: 16 passes of prefetch would spill)
    uint4 res_v[PREFETCH ? PASSES : 1], aux_v[PREFETCH ? PASSES : 1];
    const bool want_aux = (g.epi == EPI_DGELU || g.epi == EPI_DGELU_ONLY || g.epi == EPI_DRELU || g.epi == EPI_MUL_AUX);
    const float alpha = (g.alpha_dev ? g.alpha * *g.alpha_dev : g.alpha) * (g.alpha_dev2 ? *g.alpha_dev2 : 1.0f);
#pragma unroll
    for (int p = 0; p < (PREFETCH ? PASSES : 0); ++p) {
        const int rl = tr + (p % QP) * RPP;
        const int gr = m0 + (p / QP) * 64 + rl;
        const bool ok = col_ok && (rl < 64) && (gr < g.M);
        res_v[p] = make_uint4(0, 0, 0, 0);
        aux_v[p] = make_uint4(0, 0, 0, 0);
        if (ok && g.residual) res_v[p] = *reinterpret_cast<const uint4*>(g.residual + (size_t)gr * g.ldr + gc);
        if (ok && want_aux) aux_v[p] = *reinterpret_cast<const uint4*>(g.aux_in + (size_t)gr * g.ldai + gc);
    }
    // accumulators -> LDS -> coalesced 16-byte row segments, one 64-row slab (= one wave row) at a time.
    //  * The epilogue mode is uniform: ONE switch per tile, each arm the whole slab sequence with the mode a constant (tag = mode + 16 * residual;
    //    tag < 0: the arm for combinations no caller uses - everything read at run time, as round 2 had it for every call).
    //  * Wide tiles read a slab's residual / saved pre-activation rows one slab AHEAD (NT_EPI_READS_LATE: at the top of their own slab): the reads of
    //    slab q + 1 are in flight during the arithmetic and the stores of slab q, and vmcnt retires in order, so the wait in front of their first
    //    use leaves those stores in flight too.  Two register sets, addressed by constants (q & 1) - a copy between sets makes hipcc wait.
    //  * The slab index is a template constant (a `#pragma unroll` loop of this size is not reliably unrolled).
    float* Cs = reinterpret_cast<float*>(smem);
    auto epilogue = [&](auto epi_tag) {
        constexpr int TAG = decltype(epi_tag)::value;
        constexpr int EPI = TAG < 0 ? -1 : (TAG & 15), RES = TAG < 0 ? -1 : (TAG >> 4);
        constexpr bool AUX = (EPI == EPI_DGELU || EPI == EPI_DGELU_ONLY || EPI == EPI_DRELU || EPI == EPI_MUL_AUX);
#ifdef NT_EPI_READS_LATE
        constexpr bool AHEAD = false;
#else
        constexpr bool AHEAD = !PREFETCH && TAG >= 0 && (AUX || RES == 1);
#endif
        uint4 rs_v[AHEAD ? 2 : 1][PREFETCH ? 1 : QP], as_v[AHEAD ? 2 : 1][PREFETCH ? 1 : QP];
        auto slab_reads = [&](auto q_tag) {
            constexpr int q = decltype(q_tag)::value, S = AHEAD ? (q & 1) : 0;
#pragma unroll
            for (int hp = 0; hp < (PREFETCH ? 0 : QP); ++hp) {
                const int rl = tr + hp * RPP;
                const int gr = m0 + q * 64 + rl;
                if constexpr (TAG >= 0) {
                    // UNCONDITIONAL reads from clamped addresses (rows past the edge re-read row M - 1 and are never stored): with the reads under
                    // `if (ok)`, hipcc's wait-count pass meets joins where a destination register may still be a pending load of a skipped row and
                    // puts s_waitcnt vmcnt(0) between the slab's reads - one exposed HBM latency per row instead of one per slab.
                    const size_t grc = (size_t)min(gr, g.M - 1);
                    const int gcc = col_ok ? gc : 0;
                    if constexpr (RES == 1) rs_v[S][hp] = *reinterpret_cast<const uint4*>(g.residual + grc * g.ldr + gcc);
                    if constexpr (AUX) as_v[S][hp] = *reinterpret_cast<const uint4*>(g.aux_in + grc * g.ldai + gcc);
                } else {
                    const bool ok = col_ok && (rl < 64) && (gr < g.M);
                    rs_v[S][hp] = make_uint4(0, 0, 0, 0);
                    as_v[S][hp] = make_uint4(0, 0, 0, 0);
                    if (ok && g.residual) rs_v[S][hp] = *reinterpret_cast<const uint4*>(g.residual + (size_t)gr * g.ldr + gc);
                    if (ok && want_aux) as_v[S][hp] = *reinterpret_cast<const uint4*>(g.aux_in + (size_t)gr * g.ldai + gc);
                }
            }
        };
        auto slab = [&](auto q_tag) {
            constexpr int q = decltype(q_tag)::value, S = AHEAD ? (q & 1) : 0;
            NTP_T(nt_s0);
            if constexpr (!AHEAD) slab_reads(q_tag);
            // LDS ordering is all the two slab barriers need: a __syncthreads() also waits for every outstanding global access (vmcnt(0) -
            // on CDNA4 that counter includes STORES), i.e. for the previous slab's 16-byte output stores and for the epilogue reads in flight
#ifdef NT_EPI_SYNCTHREADS
            __syncthreads();   // fragment reads (q = 0) / the previous slab's reads are done
#else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            if (wm == q) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        *reinterpret_cast<f32x4*>(Cs + (i * 16 + li) * LDCS + wn * (BN / 2) + j * 16 + 4 * lg) = acc[i][j];
            }
#ifdef NT_EPI_SYNCTHREADS
            __syncthreads();
#else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            if constexpr (AHEAD && q + 1 < WAVES_M) slab_reads(std::integral_constant<int, q + 1>{});
            NTP_T(nt_s1);
            NTP_ADD(3, nt_s0, nt_s1);
            if (col_ok) {
#pragma unroll
                for (int hp = 0; hp < QP; ++hp) {
                    const int rl = tr + hp * RPP;
                    const int gr = m0 + q * 64 + rl;
                    if (rl < 64 && gr < g.M) {
                        const uint4 rv = PREFETCH ? res_v[PREFETCH ? q * QP + hp : 0] : rs_v[S][PREFETCH ? 0 : hp];
                        const uint4 av = PREFETCH ? aux_v[PREFETCH ? q * QP + hp : 0] : as_v[S][PREFETCH ? 0 : hp];
                        if constexpr (TAG >= 0) nt_epilogue_row<EPI, RES>(g, Cs + rl * LDCS + tc, gr, gc, bias, cs, rv, av, alpha);
                        else nt_epilogue_row_rt(g, Cs + rl * LDCS + tc, gr, gc, bias, cs, rv, av, alpha);
                    }
                }
            }
            NTP_T(nt_s3);
            NTP_ADD(5, nt_s1, nt_s3);
        };
        if constexpr (AHEAD) slab_reads(std::integral_constant<int, 0>{});
        slab(std::integral_constant<int, 0>{});
        slab(std::integral_constant<int, 1>{});
        if constexpr (WAVES_M > 2) {
            slab(std::integral_constant<int, 2>{});
            slab(std::integral_constant<int, 3>{});
        }
    };
    switch (g.epi + (g.residual ? 16 : 0)) {
        case EPI_NONE: epilogue(std::integral_constant<int, EPI_NONE>{}); break;
        case EPI_NONE + 16: epilogue(std::integral_constant<int, EPI_NONE + 16>{}); break;
        case EPI_GELU: epilogue(std::integral_constant<int, EPI_GELU>{}); break;
        case EPI_DGELU: epilogue(std::integral_constant<int, EPI_DGELU>{}); break;
        case EPI_RELU: epilogue(std::integral_constant<int, EPI_RELU>{}); break;
        case EPI_RELU + 16: epilogue(std::integral_constant<int, EPI_RELU + 16>{}); break;
        case EPI_DRELU: epilogue(std::integral_constant<int, EPI_DRELU>{}); break;
        case EPI_DGELU_ONLY: epilogue(std::integral_constant<int, EPI_DGELU_ONLY>{}); break;
        case EPI_GELU_DAUX: epilogue(std::integral_constant<int, EPI_GELU_DAUX>{}); break;
        case EPI_MUL_AUX: epilogue(std::integral_constant<int, EPI_MUL_AUX>{}); break;
        default: epilogue(std::integral_constant<int, -1>{}); break;
    }
#ifdef NT_PROBE
    {
        NTP_T(nt_t3);
        ntp[0] = nt_t3 - nt_t0; ntp[6] = 1;
        if (tid == 0 && blockIdx.x < NT_PROBE_SLOTS)                 // wave 0 of the first 32768 workgroups
            for (int i = 0; i < 8; ++i) g_nt_probe[blockIdx.x][i] = (unsigned long long)ntp[i];
    }
#endif
}

template <int BM, int BN, int BK, int WAVES_M, int NST, int F8 = 0>
static void launch_nt(GemmNT& g, hipStream_t stream) {
    g.tiles_m = cdiv(g.M, BM);
    g.tiles_n = cdiv(g.N, BN);
    const size_t stage = (size_t)NST * (BM * BK * 2 + BN * BK * 2);
    const size_t cs = (size_t)64 * (BN + 4) * 4;
    const size_t shm = stage > cs ? stage : cs;
    mmg_allow_lds(gemm_nt_kernel<BM, BN, BK, WAVES_M, NST, F8>, shm);
    MMG_NOTE_KERNEL("gemm_nt_kernel<%d, %d, %d, %d, %d, %d>", BM, BN, BK, WAVES_M, NST, F8);
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, BK, WAVES_M, NST, F8>), dim3(g.tiles_m * g.tiles_n), dim3(WAVES_M * 128), shm,
                       stream, g);
}

MMG_API int mmg_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                             const float* bias, const float* colscale, const void* residual, int ldr,
                             const void* aux_in, int ldai, void* aux_out, int ldao, int epi, int out_f32, float alpha,
                             hipStream_t stream) {
    MMG_CHECK_ARG(A && B && C, "mmg_gemm_nt_bf16: null operand");
    MMG_CHECK_ARG(M > 0 && N > 0 && K > 0, "mmg_gemm_nt_bf16: M=%d N=%d K=%d must be positive", M, N, K);
    MMG_CHECK_ARG(K % 32 == 0, "mmg_gemm_nt_bf16: K=%d must be a multiple of 32", K);
    MMG_CHECK_ARG(N % 8 == 0, "mmg_gemm_nt_bf16: N=%d must be a multiple of 8", N);
    MMG_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0,
                  "mmg_gemm_nt_bf16: leading dimensions must cover the row and be multiples of 8 (lda=%d ldb=%d ldc=%d)",
                  lda, ldb, ldc);
    MMG_CHECK_ARG(epi >= EPI_NONE && epi <= EPI_MUL_AUX, "mmg_gemm_nt_bf16: unknown epilogue %d", epi);
    MMG_CHECK_ARG(!(epi == EPI_DGELU || epi == EPI_DGELU_ONLY || epi == EPI_DRELU || epi == EPI_MUL_AUX) || (aux_in && ldai >= N && ldai % 8 == 0),
                  "mmg_gemm_nt_bf16: activation-gradient epilogue needs aux_in");
    MMG_CHECK_ARG(!residual || (ldr >= N && ldr % 8 == 0), "mmg_gemm_nt_bf16: bad ldr=%d", ldr);
    MMG_CHECK_ARG(!aux_out || (ldao >= N && ldao % 8 == 0), "mmg_gemm_nt_bf16: bad ldao=%d", ldao);
    GemmNT g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.C = C; g.ldc = ldc; g.out_f32 = out_f32; g.bias = bias; g.colscale = colscale;
    g.residual = (const bf16_t*)residual; g.ldr = ldr; g.aux_in = (const bf16_t*)aux_in; g.ldai = ldai;
    g.aux_out = (bf16_t*)aux_out; g.ldao = ldao; g.epi = epi; g.alpha = alpha; g.out_fp8 = 0; g.alpha_dev = nullptr; g.alpha_dev2 = nullptr;
    static const int force_bk = getenv("MMG_GEMM_BK") ? atoi(getenv("MMG_GEMM_BK")) : 0;   // tuning knobs
    static const int use_big = getenv("MMG_GEMM_V2") ? atoi(getenv("MMG_GEMM_V2")) : 1;
    // outputs larger than the 256 MiB Infinity Cache cannot be re-read from cache anyway: stream them past L2
    // (measured: -14...-20 % on the write-heavy GELU / residual epilogues); MMG_GEMM_NT_STORE=0/1 forces it off/on
    static const int nt_mode = getenv("MMG_GEMM_NT_STORE") ? atoi(getenv("MMG_GEMM_NT_STORE")) : -1;
    g.nt_store = nt_mode >= 0 ? nt_mode : ((size_t)M * N * (out_f32 ? 4 : 2) >= ((size_t)256 << 20));
    const bool k64 = (K % 64 == 0) && force_bk != 32;
    // N tile: 96 when it divides N and 128 does not (ConvNeXt widths 96/192), else 128
    const bool n96 = (N % 128 != 0) && (N % 96 == 0);
    // K < 384 (ConvNeXt stages 1-2, stem): HBM/latency bound -> 16 KiB stages, three of them, three workgroups per CU
    static const int use_3wg = getenv("MMG_GEMM_3WG") ? atoi(getenv("MMG_GEMM_3WG")) : 1;
    // (round 2, tools/nt_knobs.py, profiles/r02_nt_tile_rules.txt: the 4.2 M x 384 x 192 data gradient of the first downsample layer ran
    // 30 % faster on the two-stage 128 x 128 x 64 tile than on the three-workgroup one, the N = 384 long-K shapes 2-4 % faster on it than
    // on 256 x 128; BERT's shapes do not care)
    static const int k3_max = getenv("MMG_GEMM_K3") ? atoi(getenv("MMG_GEMM_K3")) : 128;       // 3-WG config below this K
    static const int kbig_min = getenv("MMG_GEMM_KBIG") ? atoi(getenv("MMG_GEMM_KBIG")) : 4096; // 256x128 config from this K
    // 256x256 tile (8 waves, one workgroup per CU): 128 FLOP per operand byte pulled from L2, which is what bounds the
    // 128-wide tiles (~10 TB/s of L2->LDS traffic); used from this K upwards when N is a multiple of 256 (0 = never)
    static const int use_256 = getenv("MMG_GEMM_256") ? atoi(getenv("MMG_GEMM_256")) : 384;
    // one workgroup per CU for the 256-row tiles: a grid that fills the last round of 256 CUs badly (BERT's ~10 k packed
    // tokens x N = 768: 123 tiles = 48 % of one round) goes to the next smaller tile when that one fills better
    static const int fill_rule = getenv("MMG_GEMM_FILL") ? atoi(getenv("MMG_GEMM_FILL")) : 1;
    auto fill = [](long wgs) { return (double)wgs / (double)(cdiv(wgs, 256) * 256L); };
    const double f256 = fill((long)cdiv(M, 256) * cdiv(N, 256)), f128 = fill((long)cdiv(M, 256) * cdiv(N, 128));
    const bool fills = !fill_rule || f256 >= 0.8 * f128;
    // (round 3, tools/nt_deep_ab.sh: the same tile on 32-column stages, three or four of them - more K tiles in flight at short K - was 2-3 %
    // SLOWER on the K = 384 / 768 fat-epilogue shapes, 2883 / 2915 against 2827 us: the main loop is not waiting for its operands.  Removed.)
    // round 4: N = 192 / 384 (ConvNeXt stage-3 d LN-out = dh W1: 1 M x 384 x 1536, 13.6 ms of a C2 step on 128 x 128 tiles at 0.33 of the MFMA peak)
    // take a 256 x 192 tile - the 256 x 256 kernel's 8 waves with 64 x 96 wave tiles (0.42 fragment reads per MFMA against 0.5)
    const int use_192 = getenv("MMG_GEMM_192") ? atoi(getenv("MMG_GEMM_192")) : 384;      // (read per call: same-process A/B, tools/nt_192_ab.py)
    const double f192 = fill((long)cdiv(M, 256) * cdiv(N, 192));
    if (use_256 && k64 && N % 256 == 0 && M >= 4096 && K >= use_256 && fills) launch_nt<256, 256, 64, 4, 2>(g, stream);
    else if (use_192 && k64 && N % 192 == 0 && M >= 4096 && K >= use_192 && (!fill_rule || f192 >= 0.8 * f128)) launch_nt<256, 192, 64, 4, 2>(g, stream);
    else if (use_3wg && !n96 && K % 32 == 0 && K < k3_max) launch_nt<128, 128, 32, 2, 3>(g, stream);
    else if (use_big && k64 && !n96 && M >= 4096 && K >= kbig_min) launch_nt<256, 128, 64, 4, 3>(g, stream);
    else if (n96) { if (k64) launch_nt<128, 96, 64, 2, 2>(g, stream); else launch_nt<128, 96, 32, 2, 2>(g, stream); }
    else          { if (k64) launch_nt<128, 128, 64, 2, 2>(g, stream); else launch_nt<128, 128, 32, 2, 2>(g, stream); }
    MMG_LAUNCH_CHECK("mmg_gemm_nt_bf16");
    return 0;
}

// e4m3 x e4m3 -> fp32 accumulate on the double-rate K = 128 MFMA; same tiles, staging and epilogues as the bf16 kernel
// (an e4m3 row of K bytes is staged as a "bf16" row of K/2 elements).
MMG_API int mmg_gemm_nt_fp8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                            const float* bias, const float* colscale, const void* residual, int ldr, void* aux_out,
                            int ldao, int epi, int out_kind, float alpha, const float* alpha_dev, hipStream_t stream) {
    MMG_CHECK_ARG(A && B && C, "mmg_gemm_nt_fp8: null operand");
    MMG_CHECK_ARG(M > 0 && N > 0 && K > 0, "mmg_gemm_nt_fp8: M=%d N=%d K=%d must be positive", M, N, K);
    MMG_CHECK_ARG(K % 128 == 0, "mmg_gemm_nt_fp8: K=%d must be a multiple of 128 (one MFMA k-step)", K);
    MMG_CHECK_ARG(N % 8 == 0, "mmg_gemm_nt_fp8: N=%d must be a multiple of 8", N);
    MMG_CHECK_ARG(out_kind >= 0 && out_kind <= 2, "mmg_gemm_nt_fp8: out_kind=%d (0 bf16, 1 fp32, 2 e4m3)", out_kind);
    MMG_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && lda % 16 == 0 && ldb % 16 == 0 && ldc % 8 == 0,
                  "mmg_gemm_nt_fp8: leading dimensions must cover the row; lda/ldb multiples of 16 bytes, ldc of 8 "
                  "(lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    MMG_CHECK_ARG(epi == EPI_NONE || epi == EPI_GELU || epi == EPI_RELU || epi == EPI_GELU_DAUX, "mmg_gemm_nt_fp8: epilogue %d not available", epi);
    MMG_CHECK_ARG(!residual || (ldr >= N && ldr % 8 == 0), "mmg_gemm_nt_fp8: bad ldr=%d", ldr);
    MMG_CHECK_ARG(!aux_out || (ldao >= N && ldao % 8 == 0), "mmg_gemm_nt_fp8: bad ldao=%d", ldao);
    GemmNT g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.M = M; g.N = N; g.K = K / 2; g.lda = lda / 2; g.ldb = ldb / 2;
    g.C = C; g.ldc = ldc; g.out_f32 = out_kind == 1; g.out_fp8 = out_kind == 2; g.bias = bias; g.colscale = colscale;
    g.residual = (const bf16_t*)residual; g.ldr = ldr; g.aux_in = nullptr; g.ldai = 0;
    g.aux_out = (bf16_t*)aux_out; g.ldao = ldao; g.epi = epi; g.alpha = alpha; g.alpha_dev = alpha_dev; g.alpha_dev2 = nullptr;
    g.nt_store = (size_t)M * N * (out_kind == 1 ? 4 : out_kind == 2 ? 1 : 2) >= ((size_t)256 << 20);
    static const int tile = getenv("MMG_FP8_TILE") ? atoi(getenv("MMG_FP8_TILE")) : 0;   // tuning: 1 = 256x128, 2 = 128x128
    if (tile == 0 && N % 256 == 0 && M >= 4096) launch_nt<256, 256, 64, 4, 2, 1>(g, stream);
    else if (tile != 2 && M >= 4096) launch_nt<256, 128, 64, 4, 3, 1>(g, stream);
    else launch_nt<128, 128, 64, 2, 2, 1>(g, stream);
    MMG_LAUNCH_CHECK("mmg_gemm_nt_fp8");
    return 0;
}

// Data-gradient GEMMs of the fp8 backward (round 4): A = a gradient in OCP e5m2 (a_e5m2 != 0) or e4m3 bytes, B = e4m3 weights, fp32 accumulate on the
// K = 128 MFMA; epilogue 0 (none), 5 (x GELU'(aux_in)) or 7 (x aux_in); C bf16 / fp32 / e5m2 (out_kind 0 / 1 / 3: the gradient handed on in 8 bits,
// written once).  alpha_dev / alpha_dev2: device scalars multiplied into alpha (1 / weight scale, 1 / gradient scale).
MMG_API int mmg_gemm_nt_fp8_bwd(const void* A, int lda, int a_e5m2, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                const void* aux_in, int ldai, int epi, int out_kind, float alpha, const float* alpha_dev,
                                const float* alpha_dev2, hipStream_t stream) {
    MMG_CHECK_ARG(A && B && C, "mmg_gemm_nt_fp8_bwd: null operand");
    MMG_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 128 == 0 && N % 8 == 0, "mmg_gemm_nt_fp8_bwd: M=%d N=%d K=%d (K a multiple of 128, N of 8)", M, N, K);
    MMG_CHECK_ARG(out_kind == 0 || out_kind == 1 || out_kind == 3, "mmg_gemm_nt_fp8_bwd: out_kind=%d (0 bf16, 1 fp32, 3 e5m2)", out_kind);
    MMG_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && lda % 16 == 0 && ldb % 16 == 0 && ldc % 8 == 0,
                  "mmg_gemm_nt_fp8_bwd: leading dimensions must cover the row; lda/ldb multiples of 16 bytes, ldc of 8 (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    MMG_CHECK_ARG(epi == EPI_NONE || epi == EPI_DGELU_ONLY || epi == EPI_MUL_AUX, "mmg_gemm_nt_fp8_bwd: epilogue %d not available", epi);
    MMG_CHECK_ARG(epi == EPI_NONE || (aux_in && ldai >= N && ldai % 8 == 0), "mmg_gemm_nt_fp8_bwd: the activation-gradient epilogues need aux_in");
    GemmNT g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.M = M; g.N = N; g.K = K / 2; g.lda = lda / 2; g.ldb = ldb / 2;
    g.C = C; g.ldc = ldc; g.out_f32 = out_kind == 1; g.out_fp8 = out_kind == 3 ? 2 : 0; g.bias = nullptr; g.colscale = nullptr;
    g.residual = nullptr; g.ldr = 0; g.aux_in = (const bf16_t*)aux_in; g.ldai = ldai; g.aux_out = nullptr; g.ldao = 0;
    g.epi = epi; g.alpha = alpha; g.alpha_dev = alpha_dev; g.alpha_dev2 = alpha_dev2;
    g.nt_store = (size_t)M * N * (out_kind == 1 ? 4 : out_kind == 3 ? 1 : 2) >= ((size_t)256 << 20);
    if (a_e5m2) {
        if (N % 256 == 0 && M >= 4096) launch_nt<256, 256, 64, 4, 2, 2>(g, stream);
        else launch_nt<128, 128, 64, 2, 2, 2>(g, stream);
    } else {
        if (N % 256 == 0 && M >= 4096) launch_nt<256, 256, 64, 4, 2, 1>(g, stream);
        else launch_nt<128, 128, 64, 2, 2, 1>(g, stream);
    }
    MMG_LAUNCH_CHECK("mmg_gemm_nt_fp8_bwd");
    return 0;
}

// =============================================================================================
// TN (weight gradient): C[N1,N2] += A[M,N1]^T B[M,N2], fp32 atomics, reduction split over workgroups
// =============================================================================================

#define TN_T 128     // output tile edge
#define TN_BK 64     // reduction rows per stage

// physical 16-byte chunk of logical chunk c in reduction row m (row = 256 B = one LDS bank row)
__device__ __forceinline__ int tn_swz(int m, int c) {
    const int f = (m & 3) | (((m >> 3) & 1) << 2);
    return (((c >> 1) ^ f) << 1) | (c & 1);
}

// rows m (reduction) x 128 columns (col0.., clamped) -> lane-linear LDS tile, swizzled on the source side
template <int BK>
__device__ __forceinline__ void stage_tn(const bf16_t* __restrict__ G, int ld, int m0, int M, int col0, int ncols,
                                         char* lds_tile, int tid) {
#pragma unroll
    for (int it = 0; it < (BK * 16) / GEMM_THREADS; ++it) {
        const int p = it * GEMM_THREADS + tid;
        const int r = p >> 4, s = p & 15;
        // tn_swz is an involution in c for fixed m: logical chunk stored at physical slot s
        const int c = tn_swz(r, s);
        const int gm = min(m0 + r, M - 1);
        const int gc = max(min(col0 + c * 8, ncols - 8), 0);
        glds16(G + (size_t)gm * ld + gc, lds_tile + (size_t)(it * GEMM_THREADS + (tid & ~63)) * 16);
    }
}

// transposed fragment of a 16-column block: lane (i, g) receives tile[mk + 8g + 0..7][cb*16 + i]
__device__ __forceinline__ bf16x8 read_frag_tr(const char* lds_tile, int mk, int cb, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int m_a = mk + 8 * g + q, m_b = m_a + 4;
    const int c = cb * 2 + (p >> 1);   // logical 16-byte chunk holding columns cb*16 + 4p .. 4p+3
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    const char* pa = lds_tile + m_a * 256 + tn_swz(m_a, c) * 16 + (p & 1) * 8;
    const char* pb = lds_tile + m_b * 256 + tn_swz(m_b, c) * 16 + (p & 1) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)pa);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)pb);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Output tile (128*K1) x (128*K2), 2 x 2 waves, each wave (4*K1) x (4*K2) fragments; (K1, K2) in {(1,1), (1,2), (2,1)}.
// Why the 256-wide variants: every tile of the WIDE operand re-reads the whole narrow operand for its chunk of rows
// (measured: 2.5x the algorithmic bytes left the L2s, and the step as a whole runs at ~4 TB/s of HBM traffic), so halving the
// number of wide tiles removes a third of the traffic of the ConvNeXt stage-2/3 shapes.  Their stages are 32 rows (24 KiB)
// so that two independent workgroups still share a CU (one 8-wave workgroup per CU ran in lockstep on the stage barrier and
// was no faster).
template <int K1, int K2, int BK>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_tn_kernel(const GemmTN g) {
    constexpr int FM = 4 * K1, FN = 4 * K2;
    constexpr int SUB = BK * TN_T * 2;               // one 128-column image of a stage
    constexpr int STAGE = (K1 + K2) * SUB;
    constexpr int T1 = K1 * TN_T, T2 = K2 * TN_T, LDCS = T2 + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int w1 = wave >> 1, w2 = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    // Workgroup -> (tile, reduction chunk).  The hardware deals workgroups to the 8 XCDs round-robin (XCD = id % 8) and every
    // output tile (t1, t2) streams operand block t1 of A and t2 of B for its chunk of rows.  The tiles that share a block of
    // the WIDER operand for one chunk form a group; a group runs back to back on ONE XCD (its L2 then fetches that block once
    // instead of once per XCD) and consecutive groups go to consecutive XCDs.  (Putting a whole chunk on one XCD was
    // measured 1.5-3x slower: every workgroup of the XCD then hits the same L2 channels at the same time.)
    int t1, t2, chunk;
    if (g.xcd_order) {
        const int L = blockIdx.x, xcd = L & 7, i = L >> 3;
        const int G = g.xcd_order == 1 ? g.tiles1 : g.tiles2;            // group size = tiles of the narrow operand
        const int nwide = g.xcd_order == 1 ? g.tiles2 : g.tiles1;
        const int slot = i / G, m = i - slot * G, grp = slot * 8 + xcd;
        if (grp >= nwide * g.chunks) return;
        chunk = grp / nwide;
        const int wide = grp - chunk * nwide;
        if (g.xcd_order == 1) { t1 = m; t2 = wide; } else { t1 = wide; t2 = m; }
    } else {
        const int tile = blockIdx.x;
        chunk = blockIdx.y;
        t1 = tile / g.tiles2; t2 = tile - t1 * g.tiles2;
    }
    const int c1 = t1 * T1, c2 = t2 * T2;
    const int m_begin = chunk * g.rows_per_chunk;
    const int m_end = min(m_begin + g.rows_per_chunk, g.M);
    if (m_begin >= m_end) return;

    // fragments wholly beyond N1 / N2 (tile wider than the matrix) are skipped: wave-uniform predicates
    bool a_on[FM], b_on[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) a_on[i] = c1 + (w1 * FM + i) * 16 < g.N1;
#pragma unroll
    for (int j = 0; j < FN; ++j) b_on[j] = c2 + (w2 * FN + j) * 16 < g.N2;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient: one extra MFMA per A fragment against an all-ones operand gives the column sums of A in every
    // accumulator row; only the workgroups of the first N2 tile (and their w2 == 0 waves) do it.
    const bool do_colsum = (g.colsum_a != nullptr) && (t2 == 0) && (w2 == 0);
    f32x4 accb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const short one_bf = (short)0x3F80;
    const bf16x8 ones = {one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf};

    auto stage = [&](int buf, int m0) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int k = 0; k < K1; ++k) stage_tn<BK>(g.A, g.lda, m0, g.M, c1 + k * TN_T, g.N1, base + k * SUB, tid);
#pragma unroll
        for (int k = 0; k < K2; ++k) stage_tn<BK>(g.B, g.ldb, m0, g.M, c2 + k * TN_T, g.N2, base + (K1 + k) * SUB, tid);
    };

    const int nk = (m_end - m_begin + BK - 1) / BK;
    stage(0, m_begin);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        const int mt = m_begin + kt * BK;
        char* cbase = smem + cur * STAGE;
        if (mt + BK > m_end) {
            // ragged end of the reduction: rows >= m_end must contribute zero (they hold clamped duplicates)
            const int valid = m_end - mt, per = (BK - valid) * 16;
            for (int p = tid; p < per * (K1 + K2); p += GEMM_THREADS) {
                const int img = p / per, q = p - img * per;
                *reinterpret_cast<uint4*>(cbase + img * SUB + (valid + (q >> 4)) * 256 + (q & 15) * 16) = make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
        }
        if (kt + 1 < nk) stage(cur ^ 1, mt + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int cb = w1 * FM + i;
                af[i] = read_frag_tr(cbase + (cb >> 3) * SUB, ks * 32, cb & 7, lane);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int cb = w2 * FN + j;
                bfr[j] = read_frag_tr(cbase + (K1 + (cb >> 3)) * SUB, ks * 32, cb & 7, lane);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (a_on[i]) {
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        if (b_on[j]) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
            }
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < FM; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accb[i], 0, 0, 0);
            }
        }
    }
    if (do_colsum && lg == 0) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int n1 = c1 + (w1 * FM + i) * 16 + li;
            if (n1 < g.N1) atomicAdd(g.colsum_a + n1, accb[i][0] * g.alpha);
        }
    }
    // swapped issue: lane (li, lg) holds C[n1 = .. + li][n2 = .. + 4 lg + 0..3]; 64-row slabs go through LDS and leave as
    // 256-byte contiguous atomic rows (one wave instruction = 64 consecutive floats)
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int SLABS = T1 / 64, SPW = FM / 4;                 // slabs per tile, slabs per wave row
#pragma unroll
    for (int sl = 0; sl < SLABS; ++sl) {
        __syncthreads();
        if (sl / SPW == w1) {
            const int i0 = (sl % SPW) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *reinterpret_cast<f32x4*>(Cs + (i * 16 + li) * LDCS + (w2 * FN + j) * 16 + 4 * lg) = acc[i0 + i][j];
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * T2; idx += GEMM_THREADS) {
            const int r = idx / T2, c = idx - r * T2;
            const int gr = c1 + sl * 64 + r, gc = c2 + c;
            if (gr < g.N1 && gc < g.N2) atomicAdd(g.C + (size_t)gr * g.ldc + gc, Cs[r * LDCS + c] * g.alpha);
        }
    }
}

template <int K1, int K2, int BK>
static void launch_tn(GemmTN& g, hipStream_t stream) {
    static const int target_wgs = getenv("MMG_TN_WGS") ? atoi(getenv("MMG_TN_WGS")) : 512;
    static const int xcd = getenv("MMG_TN_XCD") ? atoi(getenv("MMG_TN_XCD")) : 1;
    g.tiles1 = cdiv(g.N1, K1 * TN_T);
    g.tiles2 = cdiv(g.N2, K2 * TN_T);
    const int tiles = g.tiles1 * g.tiles2;
    int chunks = target_wgs / tiles;
    if (chunks < 1) chunks = 1;
    const int max_chunks = cdiv(g.M, 64);
    if (chunks > max_chunks) chunks = max_chunks;
    g.rows_per_chunk = cdiv(cdiv(g.M, chunks), 64) * 64;
    chunks = cdiv(g.M, g.rows_per_chunk);
    g.chunks = chunks;
    g.xcd_order = xcd ? (g.N2 >= g.N1 ? 1 : 2) : 0;
    if (g.xcd_order) {       // whole groups per XCD: fall back when that would overfill an XCD's share of the workgroup budget
        const int G = g.xcd_order == 1 ? g.tiles1 : g.tiles2, ngroups = (tiles / G) * chunks;
        if (((ngroups + 7) / 8) * G * 8 > target_wgs && tiles * chunks <= target_wgs) g.xcd_order = 0;
    }
    const size_t stage = 2 * (size_t)(K1 + K2) * (BK * TN_T * 2);
    const size_t cs = (size_t)64 * (K2 * TN_T + 4) * 4;
    const size_t shm = stage > cs ? stage : cs;
    mmg_allow_lds(gemm_tn_kernel<K1, K2, BK>, shm);
    MMG_NOTE_KERNEL("gemm_tn_kernel<%d, %d, %d>", K1, K2, BK);
    if (g.xcd_order) {
        const int G = g.xcd_order == 1 ? g.tiles1 : g.tiles2, ngroups = (tiles / G) * chunks;
        hipLaunchKernelGGL((gemm_tn_kernel<K1, K2, BK>), dim3(8 * ((ngroups + 7) / 8) * G), dim3(GEMM_THREADS), shm, stream, g);
    } else {
        hipLaunchKernelGGL((gemm_tn_kernel<K1, K2, BK>), dim3(tiles, chunks), dim3(GEMM_THREADS), shm, stream, g);
    }
}

MMG_API int mmg_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                             float alpha, float* colsum_a, hipStream_t stream) {
    MMG_CHECK_ARG(A && B && C, "mmg_gemm_tn_bf16: null operand");
    MMG_CHECK_ARG(M > 0 && N1 >= 8 && N2 >= 8, "mmg_gemm_tn_bf16: M=%d N1=%d N2=%d", M, N1, N2);
    MMG_CHECK_ARG(N1 % 8 == 0 && N2 % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= N1 && ldb >= N2 && ldc >= N2,
                  "mmg_gemm_tn_bf16: N1=%d N2=%d lda=%d ldb=%d ldc=%d must be multiples of 8 and consistent", N1, N2,
                  lda, ldb, ldc);
    GemmTN g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.M = M; g.N1 = N1; g.N2 = N2; g.lda = lda; g.ldb = ldb;
    g.C = C; g.ldc = ldc; g.alpha = alpha; g.colsum_a = colsum_a;
    // 256-wide tiles on the wider side when it is a multiple of 256 (measured: -6...-19 % on the ConvNeXt shapes, slower on
    // the short BERT reductions where the tile count is what fills the GPU); MMG_TN_WIDE=0 disables
    static const int wide = getenv("MMG_TN_WIDE") ? atoi(getenv("MMG_TN_WIDE")) : 1;
    // long reductions onto few output columns (ConvNeXt weight gradients): the 8-wave wide-tile kernel (gemm_tn_wide.hip) streams
    // both operands once per 192x384-class tile; MMG_TN_WIDE8=0 keeps the kernels of this file
    const int wide8 = getenv("MMG_TN_WIDE8") ? atoi(getenv("MMG_TN_WIDE8")) : 1;
    if (wide8 && mmg_tn_wide_launch(g, stream)) {
        MMG_LAUNCH_CHECK("mmg_gemm_tn_bf16");
        return 0;
    }
    const bool long_m = M >= 32768;
    if (wide && long_m && N2 >= N1 && N2 % 256 == 0) launch_tn<1, 2, 32>(g, stream);
    else if (wide && long_m && N1 > N2 && N1 % 256 == 0) launch_tn<2, 1, 32>(g, stream);
    else launch_tn<1, 1, 64>(g, stream);
    MMG_LAUNCH_CHECK("mmg_gemm_tn_bf16");
    return 0;
}

// =============================================================================================
// column sums of a bf16 matrix (bias gradients): out[n] += sum_m A[m,n]
// =============================================================================================
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ A, int lda, int M, int N,
                                                     int rows_per_block, float* __restrict__ out) {
    extern __shared__ float sums[];   // [N]
    for (int i = threadIdx.x; i < N; i += 256) sums[i] = 0.f;
    __syncthreads();
    const int ng = N / 8;                       // 16-byte column groups
    const int lanes_r = 256 / ng > 0 ? 256 / ng : 1;
    const int m_begin = blockIdx.x * rows_per_block, m_end = min(m_begin + rows_per_block, M);
    for (int cg0 = 0; cg0 < ng; cg0 += 256) {
        const int cg = cg0 + (threadIdx.x % (ng < 256 ? ng : 256));
        const int rl = threadIdx.x / (ng < 256 ? ng : 256);
        if (cg < ng && rl < lanes_r) {
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int m = m_begin + rl; m < m_end; m += lanes_r) {
                const uint4 v = *reinterpret_cast<const uint4*>(A + (size_t)m * lda + cg * 8);
                s[0] += bf2f_lo(v.x); s[1] += bf2f_hi(v.x); s[2] += bf2f_lo(v.y); s[3] += bf2f_hi(v.y);
                s[4] += bf2f_lo(v.z); s[5] += bf2f_hi(v.z); s[6] += bf2f_lo(v.w); s[7] += bf2f_hi(v.w);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&sums[cg * 8 + e], s[e]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) atomicAdd(out + i, sums[i]);
}

MMG_API int mmg_colsum_bf16(const void* A, int lda, int M, int N, float* out, hipStream_t stream) {
    MMG_CHECK_ARG(A && out && M > 0 && N > 0 && N % 8 == 0 && lda % 8 == 0 && lda >= N && N <= 8192,
                  "mmg_colsum_bf16: bad argument (M=%d N=%d lda=%d)", M, N, lda);
    int blocks = cdiv(M, 256);
    if (blocks > 2048) blocks = 2048;
    const int rpb = cdiv(M, blocks);
    blocks = cdiv(M, rpb);
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), N * sizeof(float), stream, (const bf16_t*)A, lda, M, N, rpb, out);
    MMG_LAUNCH_CHECK("mmg_colsum_bf16");
    return 0;
}
