// Fused ConvNeXt block MLP for the narrow stages (C <= 384):   y = x + gamma * ( GELU( LN(d) W1^T + b1 ) W2^T + b2 )
//
// Replaces, for torchvision's CNBlock (reference: mmgclip/networks/encoder.py:53 runs `model.features`; module tree in
// notebooks/clf_convnext_tiny_experimental.ipynb cell 3: LayerNorm -> Linear(C,4C) -> GELU -> Linear(4C,C) -> layer_scale
// -> residual), the LayerNorm kernel + two GEMM launches whose 4C-wide intermediates made stages 1-2 HBM-bound
// (17C bytes/pixel of traffic in the forward).  Here the 4C hidden row never leaves the CU: HBM traffic is d (C) in, x (C)
// in, y (C) out.
//
// Structure (one workgroup = 4 waves x 32 rows; 8 waves x 16 rows at C = 384 - MlpCfg; each wave owns its rows end to end):
//   * the wave loads its rows of d straight into MFMA B-operand fragments (lane = row, 8 consecutive k), LayerNorm is done
//     in registers (row statistics need two xor-shuffles: the 4 lanes {l, l^16, l^32, l^48} share a row);
//   * the two weight matrices are pre-packed on the device (mmg_cnblock_pack_weights) into per-chunk LDS images, so a
//     chunk (NC hidden columns: NC x C of W1 and C x NC of W2) is one linear 16-byte global_load_lds stream, double
//     buffered, one barrier per chunk; fragment reads are conflict-free ds_read_b128 by construction (16 consecutive
//     granules per 16-lane group);
//   * both MFMAs are issued swapped (A := weight fragment), so the accumulator of GEMM 1 leaves lane l with row l&15 and 4
//     consecutive row positions of the weight tile: bias + GELU are applied in place and two such tiles ARE the B operand
//     of GEMM 2's 16x16x32 step - no LDS round trip for the hidden row.  The packing places the hidden units (and the output
//     columns) so that a lane's pair of tiles is 8 consecutive units: every global access of the kernels is 16 bytes.
// The bound is the fp32 VALU (erf-GELU: the 11-operation polynomial form of common.h, gelu_bf16 - its result is rounded to bf16
// right away; still ~2x the MFMA time at C=96), then HBM; the
// backward (cnblock_mlp_bwd_kernel below) recomputes the hidden row and is bound by its 4C-wide stores.
#include "common.h"
#include <stdlib.h>


struct MlpFwd {
    const bf16_t* xd;                       // [M,C] depthwise output
    const float* ln_w; const float* ln_b; float eps;
    const bf16_t* wt;                       // packed chunks
    const float* b1; const float* b2; const float* gamma;
    const bf16_t* res;                      // [M,C] block input
    bf16_t* y;                              // [M,C]
    bf16_t* hpre;                           // optional [M,4C] pre-GELU hidden (for an unfused backward)
    bf16_t* xln;                            // optional [M,C] LayerNorm output (operand of that backward's weight-gradient GEMM)
    float* mean; float* rstd;               // optional [M]
    long M; int ntiles;
    int nt;                                 // 4C-wide output larger than the Infinity Cache: store it past L2 (nontemporal)
    bf16_t* gact;                           // optional [M,4C] GELU(hidden) as the second GEMM consumed it (operand of that backward's dW2 GEMM)
    int hkind;                              // what `hpre` receives: 0 = the pre-GELU hidden, 1 = GELU'(hidden) (round 4: that backward's data-gradient
                                            // GEMM then runs epilogue 7, one multiply per element)
};

__device__ __forceinline__ void mlp_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Work split per width.  Up to C = 256 a wave owns 32 rows (two 16-row MFMA tiles share every weight fragment read); at C = 384
// the per-row state (C/4 operand + C/4 accumulator registers per tile) only leaves room for one tile per wave, so the
// workgroup is 8 waves of 16 rows (one workgroup per CU) and the kernel is LDS-read bound (one fragment read per MFMA).
template <int C> struct MlpCfg {
    static constexpr int NC = (C <= 128) ? 64 : 32;                     // hidden columns per streamed chunk
    static constexpr int MT = (C <= 256) ? 2 : 1;                       // 16-row tiles per wave
    static constexpr int WAVES = (C <= 256) ? 4 : 8;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int BM = WAVES * 16 * MT;                          // rows per workgroup tile
    static constexpr int WGS = (C <= 96) ? 3 : (C <= 192 ? 2 : 1);      // forward workgroups per CU (registers / LDS)
    static constexpr int WGS_BWD = (C <= 192) ? 2 : 1;
};

// ---- weight packing --------------------------------------------------------------------------------------------------
// chunk j (hidden columns j*NC .. +NC) is a sequence of parts, each an LDS image of NC*C bf16:
//   A-type part : (C/8) k-granules x NC rows x 8   element (n, c), K = c           (weight fragment of  . x [*, C]^T)
//   B-type part : (NC/8) k-granules x C rows x 8   element (c, n), K = n PERMUTED  (weight fragment of  hidden x [C, *]^T)
// Hidden-unit order: after the swapped MFMA a lane (row li, group q = lane >> 4) holds 4 consecutive ROW POSITIONS 4q..4q+3 of
// each of the two 16-row weight tiles of a 32-unit group.  The A-type image places hidden unit 8q + 4tt + r at position
// (tile tt, 4q + r), so the lane's two tiles are the 8 CONSECUTIVE units 8q..8q+7: one 16-byte store / load per lane for the
// 4C-wide tensors (8-byte pieces were bound by the texture-address path), and the k slot 8q + j that the pair of tiles
// presents as a B operand is hidden unit 8q + j, i.e. the K order of B-type parts is natural.  The ROWS of a B-type part
// (output columns c of the C-wide results) follow the same rule, so those leave as 16-byte pieces too.
// A source is W1-like ([4C, C], element (n, c) at n*C + c) or W2-like ([C, 4C], element (n, c) at c*4C + n); `scale` ([C],
// optional) multiplies by scale[c] (folds the layer scale into W2^T for the backward).
struct PackPart { const float* src; int w2_like; const float* scale; int btype; };
struct PackArgs { PackPart part[3]; int parts; bf16_t* out; int C, NC; };

__global__ __launch_bounds__(256) void mlp_pack_kernel(const PackArgs a) {
    const int C = a.C, NC = a.NC, H4 = 4 * C;
    const long per_part = (long)NC * C, per_chunk = per_part * a.parts, total = per_chunk * (H4 / NC);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e / per_chunk);
        long off = e - (long)j * per_chunk;
        const int pi = (int)(off / per_part);
        off -= (long)pi * per_part;
        const PackPart pp = a.part[pi];
        int n, c;
        if (!pp.btype) {
            const int kg = (int)(off / (NC * 8)), nn = (int)((off / 8) % NC), jj = (int)(off % 8);
            // row position nn of the image = 32-column group, MFMA tile tt, lane group q, register r  ->  hidden unit 8q + 4tt + r
            const int g32 = nn >> 5, tt = (nn >> 4) & 1, q = (nn >> 2) & 3, r = nn & 3;
            n = j * NC + 32 * g32 + 8 * q + 4 * tt + r; c = 8 * kg + jj;
        } else {
            const int kg = (int)(off / (C * 8)), jj = (int)(off % 8);
            const int cp = (int)((off / 8) % C);  // row position -> output column, same rule: the lane's pair of tiles = 8 columns
            c = (cp & ~31) + 8 * ((cp >> 2) & 3) + 4 * ((cp >> 4) & 1) + (cp & 3);
            n = j * NC + 8 * kg + jj;             // k slot 8q + j  <->  hidden unit 8q + j (see above): natural order
        }
        float v = pp.w2_like ? pp.src[(long)c * H4 + n] : pp.src[(long)n * C + c];
        if (pp.scale) v *= pp.scale[c];
        a.out[e] = f2bf(v);
    }
}

#ifdef MLP_PROBE
// Debug builds only (tools/mlp_probe.py): per-phase shader-clock totals of the forward, summed over waves.
//   [0] whole tile loop  [1] chunk-top wait + barrier  [2] GEMM 1  [3] GELU  [4] GEMM 2  [5] waves
__device__ unsigned long long g_mlp_probe[8];
MMG_API int mmg_debug_mlp_probe(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mlp_probe), 64) != hipSuccess) return 1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_probe), z, 64) != hipSuccess) return 1;
    return 0;
}
#define PROBE_T(var) const long long var = __builtin_readcyclecounter()
#define PROBE_ADD(acc, a, b) acc += (b) - (a)
#else
#define PROBE_T(var)
#define PROBE_ADD(acc, a, b)
#endif

// ---- forward ---------------------------------------------------------------------------------------------------------
// RW > 0 (round 4, C = 96 only: both weight matrices are 2 x 72 KiB = 144 KiB, they FIT the CU's 160 KiB of LDS): the "resident"
// form.  ONE workgroup of RW waves per CU stages all 4C / NC chunk images once, in its prologue; after that single barrier the waves
// never meet again - no per-chunk DMA stream (the streaming form moves the whole 144 KiB from L2 into LDS for every 128 rows: two
// thirds of the bytes entering the CU), no per-chunk `vmcnt(0)` + barrier, and each wave walks its own 32-row tiles (tile index by
// wave, not by workgroup), so the waves of a SIMD drift apart and one wave's GELU block runs beside another's MFMAs.
template <int C, bool SAVE, int RW = 0>
__global__ __launch_bounds__(RW ? RW * 64 : MlpCfg<C>::THREADS, RW ? RW / 4 : MlpCfg<C>::WGS) void cnblock_mlp_fwd_kernel(const MlpFwd p) {
    constexpr bool RES = RW > 0;
    constexpr int NC = MlpCfg<C>::NC, MT = MlpCfg<C>::MT, MLP_THREADS = RES ? RW * 64 : MlpCfg<C>::THREADS;
    constexpr int MLP_BM = RES ? 16 * MT : MlpCfg<C>::BM;        // (resident form: `tile` counts 32-row WAVE tiles)
    constexpr int KS1 = C / 32, CT = C / 16, NSUB = NC / 32, NCH = 4 * C / NC;
    constexpr int PART = NC * C * 2, CHUNK = 2 * PART, LOADS = CHUNK / 16 / MLP_THREADS;
    static_assert(CHUNK % (16 * MLP_THREADS) == 0, "chunk must be a whole number of 16-byte granules per thread");
    static_assert(NCH % 2 == 0, "ring parity is carried across tiles");
    static_assert(!RES || (!SAVE && NCH * CHUNK + 8 * C * 4 <= 160 * 1024), "resident form: every chunk image in LDS, nothing saved");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_lnw = reinterpret_cast<float*>(smem + (RES ? NCH : 2) * CHUNK);   // [C] [C] [4C] [C] [C]
    float* s_lnb = s_lnw + C;
    float* s_b1 = s_lnb + C;
    float* s_b2 = s_b1 + 4 * C;
    float* s_gm = s_b2 + C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = RES ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;      // (resident form: the tile index is per wave - keep it scalar)
    const int li = lane & 15, lg = lane >> 4;
    char* lds_wave = smem + (tid & ~63) * 16;
    const char* wsrc = reinterpret_cast<const char*>(p.wt) + tid * 16;

    auto stage = [&](int buf, int ch) {
        const char* s = wsrc + (size_t)ch * CHUNK;
#pragma unroll
        for (int it = 0; it < LOADS; ++it) mlp_glds16(s + it * (MLP_THREADS * 16), lds_wave + buf * CHUNK + it * (MLP_THREADS * 16));
    };
    if constexpr (RES) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) stage(ch, ch);
    } else if ((int)blockIdx.x < p.ntiles) stage(0, 0);
    for (int i = tid; i < C; i += MLP_THREADS) {
        s_lnw[i] = p.ln_w[i]; s_lnb[i] = p.ln_b[i]; s_b2[i] = p.b2[i]; s_gm[i] = p.gamma[i];
    }
    for (int i = tid; i < 4 * C; i += MLP_THREADS) s_b1[i] = p.b1[i];
    if constexpr (RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef MLP_PROBE
    long long pr_wait = 0, pr_g1 = 0, pr_gelu = 0, pr_g2 = 0;
#endif
    PROBE_T(pr_start);
    for (int tile = RES ? (int)blockIdx.x * RW + wave : (int)blockIdx.x; tile < p.ntiles; tile += RES ? (int)gridDim.x * RW : (int)gridDim.x) {
        const long row0 = RES ? (long)tile * MLP_BM + li : (long)tile * MLP_BM + wave * (16 * MT) + li;          // + 16*mi
        // ---- rows of d -> LayerNorm -> bf16 B-operand fragments ---------------------------------------------------------
        bf16x8 xf[MT][KS1];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const long row = row0 + 16 * mi;
            const long rr = row < p.M ? row : p.M - 1;
            uint4 raw[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) raw[ks] = *reinterpret_cast<const uint4*>(p.xd + rr * C + 32 * ks + 8 * lg);
            float v[KS1][8];
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const unsigned w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[ks][2 * e] = bf2f_lo(w[e]); v[ks][2 * e + 1] = bf2f_hi(w[e]);
                    s += v[ks][2 * e] + v[ks][2 * e + 1];
                }
            }
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            const float mean = s * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mean; q = fmaf(d, d, q); }
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            const float rstd = rsqrtf(q * (1.0f / C) + p.eps);
            if (SAVE && lg == 0 && row < p.M) { p.mean[row] = mean; p.rstd[row] = rstd; }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_lnw + 32 * ks + 8 * lg);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(s_lnw + 32 * ks + 8 * lg + 4);
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(s_lnb + 32 * ks + 8 * lg);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(s_lnb + 32 * ks + 8 * lg + 4);
                const float g[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                const float b[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                unsigned o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack2bf(fmaf((v[ks][2 * e] - mean) * rstd, g[2 * e], b[2 * e]),
                                   fmaf((v[ks][2 * e + 1] - mean) * rstd, g[2 * e + 1], b[2 * e + 1]));
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                xf[mi][ks] = __builtin_bit_cast(bf16x8, (u32x4_t{o[0], o[1], o[2], o[3]}));
                if (SAVE && p.xln && row < p.M) *reinterpret_cast<uint4*>(p.xln + row * C + 32 * ks + 8 * lg) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }

        f32x4 yacc[MT][CT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) yacc[mi][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- hidden chunks ---------------------------------------------------------------------------------------------
        for (int ch = 0; ch < NCH; ++ch) {
            PROBE_T(pr_a);
            if constexpr (!RES) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            PROBE_T(pr_b);
            PROBE_ADD(pr_wait, pr_a, pr_b);
            if constexpr (!RES) {
                if (ch + 1 < NCH) stage((ch + 1) & 1, ch + 1);
                else if (tile + (int)gridDim.x < p.ntiles) stage(0, 0);
            }
            const char* wb = smem + (RES ? ch : (ch & 1)) * CHUNK;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
#if defined(MLP_PROBE) && MLP_PROBE >= 2
                PROBE_T(pr_c);
#endif
                // accumulators start from the bias (lane owns 4 consecutive hidden columns of its rows)
                const int n0 = ch * NC + sub * 32 + 8 * lg;        // this lane's 8 consecutive hidden units
                const f32x4 bia0 = *reinterpret_cast<const f32x4*>(s_b1 + n0);
                const f32x4 bia1 = *reinterpret_cast<const f32x4*>(s_b1 + n0 + 4);
                f32x4 hacc[MT][2];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) { hacc[mi][0] = bia0; hacc[mi][1] = bia1; }
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + (((4 * ks + lg) * NC + (2 * sub + tt) * 16 + li) << 4));
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi)
                            hacc[mi][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[mi][ks], hacc[mi][tt], 0, 0, 0);
                    }
                }
#if defined(MLP_PROBE) && MLP_PROBE >= 2
                asm volatile("" : "+v"(hacc[0][0]), "+v"(hacc[MT - 1][1]));
                PROBE_T(pr_d);
                PROBE_ADD(pr_g1, pr_c, pr_d);
#endif
                bf16x8 gf[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const f32x4 h0 = hacc[mi][0], h1 = hacc[mi][1];
                    if (SAVE) {
                        const long row = row0 + 16 * mi;
                        if (row < p.M) {
                            if (p.hkind)           // (uniform) GELU'(h) from the fp32 hidden: the exp-free polynomial the backward would have evaluated on bf16(h)
                                store16_stream(p.hpre + row * (4 * C) + n0,
                                               make_uint4(pack2bf(gelu_bf16_grad_poly(h0[0]), gelu_bf16_grad_poly(h0[1])), pack2bf(gelu_bf16_grad_poly(h0[2]), gelu_bf16_grad_poly(h0[3])),
                                                          pack2bf(gelu_bf16_grad_poly(h1[0]), gelu_bf16_grad_poly(h1[1])), pack2bf(gelu_bf16_grad_poly(h1[2]), gelu_bf16_grad_poly(h1[3]))), p.nt);
                            else
                                store16_stream(p.hpre + row * (4 * C) + n0,
                                               make_uint4(pack2bf(h0[0], h0[1]), pack2bf(h0[2], h0[3]), pack2bf(h1[0], h1[1]), pack2bf(h1[2], h1[3])), p.nt);
                        }
                    }
                    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                    gf[mi] = __builtin_bit_cast(bf16x8, (u32x4_t{pack2bf(gelu_bf16(h0[0]), gelu_bf16(h0[1])), pack2bf(gelu_bf16(h0[2]), gelu_bf16(h0[3])),
                                                                 pack2bf(gelu_bf16(h1[0]), gelu_bf16(h1[1])), pack2bf(gelu_bf16(h1[2]), gelu_bf16(h1[3]))}));
                    if (SAVE) {
                        const long row = row0 + 16 * mi;
                        if (p.gact && row < p.M) {
                            const u32x4_t gw = __builtin_bit_cast(u32x4_t, gf[mi]);
                            store16_stream(p.gact + row * (4 * C) + n0, make_uint4(gw[0], gw[1], gw[2], gw[3]), p.nt);
                        }
                    }
                }
#if defined(MLP_PROBE) && MLP_PROBE >= 2
                asm volatile("" : "+v"(gf[0]), "+v"(gf[MT - 1]));
                PROBE_T(pr_e);
                PROBE_ADD(pr_gelu, pr_d, pr_e);
#endif
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + PART + (((4 * sub + lg) * C + ct * 16 + li) << 4));
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        yacc[mi][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, gf[mi], yacc[mi][ct], 0, 0, 0);
                }
#if defined(MLP_PROBE) && MLP_PROBE >= 2
                asm volatile("" : "+v"(yacc[0][0]), "+v"(yacc[MT - 1][CT - 1]));
                PROBE_T(pr_f);
                PROBE_ADD(pr_g2, pr_e, pr_f);
#endif
            }
        }

        // ---- y = x + gamma * (acc + b2) -----------------------------------------------------------------------------------
        // (lane id from the hardware here, opaque to the optimiser: derived from `lane`, the per-lane base pointers "tensor + 16 lg bytes" are
        //  loop-invariant, hipcc hoisted them out of the tile loop, spilled them at 168 registers (resident form) and re-loaded them in every tile
        //  behind `s_waitcnt vmcnt(0)` - in front of the residual loads and again in front of the stores; round 4, ISA read)
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int lg_e = lane_e >> 4;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const long row = row0 + 16 * mi;
            if (row < p.M) {
                // c-tile pair (2j, 2j+1) of this lane = columns 32j + 8 lg .. +7
                uint4 rv[CT / 2];
#pragma unroll
                for (int j = 0; j < CT / 2; ++j) rv[j] = *reinterpret_cast<const uint4*>(p.res + row * C + 32 * j + 8 * lg_e);
#pragma unroll
                for (int j = 0; j < CT / 2; ++j) {
                    const int c8 = 32 * j + 8 * lg_e;
                    const unsigned rw[4] = {rv[j].x, rv[j].y, rv[j].z, rv[j].w};
                    unsigned o[4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 b2 = *reinterpret_cast<const f32x4*>(s_b2 + c8 + 4 * h);
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(s_gm + c8 + 4 * h);
                        const f32x4 a = yacc[mi][2 * j + h];
                        o[2 * h] = pack2bf(fmaf(gm[0], a[0] + b2[0], bf2f_lo(rw[2 * h])), fmaf(gm[1], a[1] + b2[1], bf2f_hi(rw[2 * h])));
                        o[2 * h + 1] = pack2bf(fmaf(gm[2], a[2] + b2[2], bf2f_lo(rw[2 * h + 1])), fmaf(gm[3], a[3] + b2[3], bf2f_hi(rw[2 * h + 1])));
                    }
                    *reinterpret_cast<uint4*>(p.y + row * C + c8) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
#ifdef MLP_PROBE
    PROBE_T(pr_end);
    if (lane == 0) {
        atomicAdd(&g_mlp_probe[0], (unsigned long long)(pr_end - pr_start)); atomicAdd(&g_mlp_probe[1], (unsigned long long)pr_wait);
        atomicAdd(&g_mlp_probe[2], (unsigned long long)pr_g1); atomicAdd(&g_mlp_probe[3], (unsigned long long)pr_gelu);
        atomicAdd(&g_mlp_probe[4], (unsigned long long)pr_g2); atomicAdd(&g_mlp_probe[5], 1ull);
    }
#endif
}

// ---- backward (data path) ---------------------------------------------------------------------------------------------
// Same skeleton with three weight parts per chunk [W1 | gamma*W2^T | W1^T]:  per 32 hidden columns the wave recomputes the
// pre-activation h = LN(d) W1^T + b1, forms dG = dy (gamma W2) next to it, applies GELU / GELU' in registers, writes
// g = GELU(h) and dh = dG GELU'(h) (bf16, the operands of the two weight-gradient GEMMs) and feeds dh straight into the
// accumulation of d LN-out = dh W1.  Nothing 4C-wide is READ from HBM: the forward saves no hidden tensor.
struct MlpBwd {
    const bf16_t* dy;                       // [M,C] gradient of the block output
    const bf16_t* xd;                       // [M,C] depthwise output (saved)
    const float* ln_w; const float* ln_b; float eps;
    const bf16_t* wt;                       // packed backward chunks
    const float* b1;
    bf16_t* dh; bf16_t* g;                  // [M,4C]
    bf16_t* xln; bf16_t* dxln;              // [M,C] LN output (operand of dW1), gradient w.r.t. the LN output
    float* mean; float* rstd;               // [M] for the LayerNorm backward
    const bf16_t* hpre;                     // [M,4C] saved pre-activation (RECOMP = false only)
    float* ln_dw; float* ln_db;             // optional [C]: fuse the LayerNorm backward (dxln then receives d loss / d xd)
    long M; int ntiles;
    int nt;                                 // g / dh larger than the Infinity Cache: store them past L2 (nontemporal)
};

// sum over the 16 lanes of a DPP row (the lanes that share the same output columns); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

// RECOMP = false (C = 384): the pre-activation is read back from the forward's `hpre` instead of being recomputed, so the
// W1 part, its MFMAs and the LN-output fragments drop out (one 16-row tile per wave cannot afford them: registers, and one
// LDS fragment read per MFMA); chunks are then [gamma*W2^T | W1^T].
template <int C, bool RECOMP, bool LNB>
__global__ __launch_bounds__(MlpCfg<C>::THREADS, MlpCfg<C>::WGS_BWD) void cnblock_mlp_bwd_kernel(const MlpBwd p) {
    constexpr int NC = MlpCfg<C>::NC, MT = MlpCfg<C>::MT, MLP_THREADS = MlpCfg<C>::THREADS, MLP_BM = MlpCfg<C>::BM;
    constexpr int KS1 = C / 32, CT = C / 16, NSUB = NC / 32, NCH = 4 * C / NC;
    constexpr int NPART = RECOMP ? 3 : 2, P_W2 = RECOMP ? 1 : 0, P_W1T = RECOMP ? 2 : 1;
    constexpr int PART = NC * C * 2, CHUNK = NPART * PART, LOADS = CHUNK / 16 / MLP_THREADS;
    static_assert(CHUNK % (16 * MLP_THREADS) == 0, "chunk must be a whole number of 16-byte granules per thread");
    static_assert(NCH % 2 == 0, "ring parity is carried across tiles");
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_lnw = reinterpret_cast<float*>(smem + 2 * CHUNK);   // [C] [C] [4C] [C] [C]
    float* s_lnb = s_lnw + C;
    float* s_b1 = s_lnb + C;
    float* s_dg = s_b1 + 4 * C;                                  // LayerNorm weight / bias gradient of this workgroup
    float* s_db = s_dg + C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (uniform: the addresses below split into a scalar base + one 32-bit lane offset)
    const int li = lane & 15, lg = lane >> 4;
    char* lds_wave = smem + (tid & ~63) * 16;
    // Addresses inside the chunk loop are "uniform 64-bit base (scalar registers) + per-thread 32-bit offset" (round 4, ISA read of the C = 192
    // instantiation at 256 VGPRs): as per-lane 64-bit pointers hipcc kept one register pair per DMA piece and per stored row tile, spilled them, and
    // re-loaded each from scratch behind `s_waitcnt vmcnt(0)` - every one of the nine DMA instructions of a chunk waited for the previous one to
    // LAND, and every pair of g / dh stores for all earlier stores to complete: the kernel that "was bound by its HBM stores" was serialised.
    // -> buffer instructions: descriptor + scalar offset + ONE 32-bit lane offset (as cnblock_bwdw.hip does), so there is nothing 64-bit per lane to keep.
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, NCH * CHUNK, 0x27000);
    const int toff = tid * 16;
    auto stage = [&](int buf, int ch) {
#pragma unroll
        for (int it = 0; it < LOADS; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(lds_wave + buf * CHUNK + it * (MLP_THREADS * 16)), 16, toff,
                                                 ch * CHUNK + it * (MLP_THREADS * 16), 0, 0);
    };
    const int soff = (li * (4 * C) + 8 * lg) * 2;      // byte offset of this lane's 16-byte piece inside a 16-row x 4C row block
    if ((int)blockIdx.x < p.ntiles) stage(0, 0);
    for (int i = tid; i < C; i += MLP_THREADS) { s_lnw[i] = p.ln_w[i]; s_lnb[i] = p.ln_b[i]; s_dg[i] = 0.f; s_db[i] = 0.f; }
    for (int i = tid; i < 4 * C; i += MLP_THREADS) s_b1[i] = p.b1[i];
    __syncthreads();

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const long row0 = (long)tile * MLP_BM + wave * (16 * MT) + li;
        // g / dh: one descriptor per tensor over THIS WAVE's 16 MT rows of the tile; rows past M lie outside its range and their stores are dropped
        const long wrow0 = (long)tile * MLP_BM + wave * (16 * MT);
        const long wrows = p.M - wrow0 < 0 ? 0 : (p.M - wrow0 < 16 * MT ? p.M - wrow0 : 16 * MT);
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(p.g + wrow0 * (4 * C)), 0, (int)(wrows * (4 * C) * 2), 0x27000);
        const __amdgpu_buffer_rsrc_t rs_dh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dh + wrow0 * (4 * C)), 0, (int)(wrows * (4 * C) * 2), 0x27000);
        bf16x8 xf[RECOMP ? MT : 1][RECOMP ? KS1 : 1], dyf[MT][KS1];
        float row_mean[MT], row_rstd[MT];
        // (lane group from the hardware for the global addresses of the tile prologue / epilogue: derived from `lane`, the per-lane bases "tensor +
        //  16 lg bytes" are loop-invariant 64-bit values that hipcc hoists, spills at 256 registers and re-loads per tile behind vmcnt(0))
        int lane_t;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
        const int lg_t = lane_t >> 4;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const long row = row0 + 16 * mi;
            const long rr = row < p.M ? row : p.M - 1;
            uint4 raw[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                raw[ks] = *reinterpret_cast<const uint4*>(p.xd + rr * C + 32 * ks + 8 * lg_t);
                const uint4 dv = *reinterpret_cast<const uint4*>(p.dy + rr * C + 32 * ks + 8 * lg_t);
                dyf[mi][ks] = __builtin_bit_cast(bf16x8, (u32x4_t{dv.x, dv.y, dv.z, dv.w}));
            }
            float v[KS1][8];
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const unsigned w[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[ks][2 * e] = bf2f_lo(w[e]); v[ks][2 * e + 1] = bf2f_hi(w[e]);
                    s += v[ks][2 * e] + v[ks][2 * e + 1];
                }
            }
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            const float mean = s * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mean; q = fmaf(d, d, q); }
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            const float rstd = rsqrtf(q * (1.0f / C) + p.eps);
            if (lg == 0 && row < p.M) { p.mean[row] = mean; p.rstd[row] = rstd; }
            row_mean[mi] = mean; row_rstd[mi] = rstd;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_lnw + 32 * ks + 8 * lg);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(s_lnw + 32 * ks + 8 * lg + 4);
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(s_lnb + 32 * ks + 8 * lg);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(s_lnb + 32 * ks + 8 * lg + 4);
                const float g[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                const float b[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                unsigned o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack2bf(fmaf((v[ks][2 * e] - mean) * rstd, g[2 * e], b[2 * e]),
                                   fmaf((v[ks][2 * e + 1] - mean) * rstd, g[2 * e + 1], b[2 * e + 1]));
                if (RECOMP) xf[RECOMP ? mi : 0][RECOMP ? ks : 0] = __builtin_bit_cast(bf16x8, (u32x4_t{o[0], o[1], o[2], o[3]}));
                if (row < p.M) *reinterpret_cast<uint4*>(p.xln + row * C + 32 * ks + 8 * lg_t) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }

        f32x4 dxacc[MT][CT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dxacc[mi][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch = 0; ch < NCH; ++ch) {
            // this chunk's weights have landed.  From the second chunk of a tile on, everything this wave issued after that DMA is the previous
            // chunk's 2 MT NSUB buffer stores of g / dh (always issued: rows past M are dropped by the descriptor's range check, not by a
            // branch) - the counter retires in order, so they may stay in flight instead of being waited for chunk by chunk
            constexpr int NST_CH = 2 * MT * NSUB;
            if (RECOMP && ch > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST_CH) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ch + 1 < NCH) stage((ch + 1) & 1, ch + 1);
            else if (tile + (int)gridDim.x < p.ntiles) stage(0, 0);
            const char* wb = smem + (ch & 1) * CHUNK;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int n0 = ch * NC + sub * 32 + 8 * lg;        // this lane's 8 consecutive hidden units
                const f32x4 bia0 = *reinterpret_cast<const f32x4*>(s_b1 + n0);
                const f32x4 bia1 = *reinterpret_cast<const f32x4*>(s_b1 + n0 + 4);
                f32x4 hacc[MT][2], gacc[MT][2];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    if (RECOMP) {
                        hacc[mi][0] = bia0; hacc[mi][1] = bia1;
                    } else {             // saved pre-activation (bias already in it): lane = row, 4 consecutive columns
                        const long row = row0 + 16 * mi;
                        const long rr = row < p.M ? row : p.M - 1;
                        const uint4 u = *reinterpret_cast<const uint4*>(p.hpre + rr * (4 * C) + n0);
                        hacc[mi][0] = f32x4{bf2f_lo(u.x), bf2f_hi(u.x), bf2f_lo(u.y), bf2f_hi(u.y)};
                        hacc[mi][1] = f32x4{bf2f_lo(u.z), bf2f_hi(u.z), bf2f_lo(u.w), bf2f_hi(u.w)};
                    }
                    gacc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; gacc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int off = ((4 * ks + lg) * NC + (2 * sub + tt) * 16 + li) << 4;
                        const bf16x8 w2f = *reinterpret_cast<const bf16x8*>(wb + P_W2 * PART + off);
                        if (RECOMP) {
                            const bf16x8 w1f = *reinterpret_cast<const bf16x8*>(wb + off);
#pragma unroll
                            for (int mi = 0; mi < MT; ++mi)
                                hacc[mi][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f, xf[RECOMP ? mi : 0][RECOMP ? ks : 0], hacc[mi][tt], 0, 0, 0);
                        }
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi)
                            gacc[mi][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f, dyf[mi][ks], gacc[mi][tt], 0, 0, 0);
                    }
                }
                bf16x8 dhf[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    unsigned gp[4], dp[4];
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        float a[4], d[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float dg;
                            gelu_bf16_both(hacc[mi][tt][e], a[e], dg);
                            d[e] = gacc[mi][tt][e] * dg;
                        }
                        gp[2 * tt] = pack2bf(a[0], a[1]); gp[2 * tt + 1] = pack2bf(a[2], a[3]);
                        dp[2 * tt] = pack2bf(d[0], d[1]); dp[2 * tt + 1] = pack2bf(d[2], d[3]);
                    }
                    dhf[mi] = __builtin_bit_cast(bf16x8, (u32x4_t{dp[0], dp[1], dp[2], dp[3]}));
                    {
                        const int vo = soff + mi * (16 * 4 * C * 2), so = (ch * NC + sub * 32) * 2;
                        if (p.nt) {            // (uniform) streamed past L2: aux bit 1 = nt
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{gp[0], gp[1], gp[2], gp[3]}, rs_g, vo, so, 2);
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{dp[0], dp[1], dp[2], dp[3]}, rs_dh, vo, so, 2);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{gp[0], gp[1], gp[2], gp[3]}, rs_g, vo, so, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{dp[0], dp[1], dp[2], dp[3]}, rs_dh, vo, so, 0);
                        }
                    }
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + P_W1T * PART + (((4 * sub + lg) * C + ct * 16 + li) << 4));
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        dxacc[mi][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, dhf[mi], dxacc[mi][ct], 0, 0, 0);
                }
            }
        }
        int lane_x;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_x));
        const int lg_x = lane_x >> 4;
        if (!LNB) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const long row = row0 + 16 * mi;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < CT / 2; ++j) {
                        const f32x4 a = dxacc[mi][2 * j], b = dxacc[mi][2 * j + 1];
                        *reinterpret_cast<uint4*>(p.dxln + row * C + 32 * j + 8 * lg_x) =
                            make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3]));
                    }
                }
            }
        } else {
            // LayerNorm backward on the accumulators (lane = row, 4 consecutive columns per 16-column tile):
            //   g = dxln * gamma,  d xd = rstd * (g - mean(g) - xhat * mean(g * xhat)),  dgamma += dxln * xhat,  dbeta += dxln
            uint2 xv[MT][CT];
            float m1[MT], m2[MT];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const long row = row0 + 16 * mi;
                const long rr = row < p.M ? row : p.M - 1;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xv[mi][ct] = *reinterpret_cast<const uint2*>(p.xd + rr * C + 32 * (ct >> 1) + 8 * lg_x + 4 * (ct & 1));
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(s_lnw + 32 * (ct >> 1) + 8 * lg + 4 * (ct & 1));
                    const float x[4] = {bf2f_lo(xv[mi][ct].x), bf2f_hi(xv[mi][ct].x), bf2f_lo(xv[mi][ct].y), bf2f_hi(xv[mi][ct].y)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (x[r] - row_mean[mi]) * row_rstd[mi];
                        const float gg = dxacc[mi][ct][r] * gm[r];
                        s1 += gg; s2 = fmaf(gg, xh, s2);
                    }
                }
                s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
                m1[mi] = s1 * (1.0f / C); m2[mi] = s2 * (1.0f / C);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int cc = 32 * (ct >> 1) + 8 * lg_x + 4 * (ct & 1);     // this lane's 4 columns of tile ct
                const f32x4 gm = *reinterpret_cast<const f32x4*>(s_lnw + cc);
                float cg[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const long row = row0 + 16 * mi;
                    const bool ok = row < p.M;
                    const float x[4] = {bf2f_lo(xv[mi][ct].x), bf2f_hi(xv[mi][ct].x), bf2f_lo(xv[mi][ct].y), bf2f_hi(xv[mi][ct].y)};
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (x[r] - row_mean[mi]) * row_rstd[mi];
                        const float dv = ok ? dxacc[mi][ct][r] : 0.f;
                        o[r] = row_rstd[mi] * (fmaf(-xh, m2[mi], dv * gm[r]) - m1[mi]);
                        cg[r] = fmaf(dv, xh, cg[r]); cb[r] += dv;
                    }
                    if (ok) *reinterpret_cast<uint2*>(p.dxln + row * C + cc) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tg = row16_sum(cg[r]), tb = row16_sum(cb[r]);
                    if (li == 0) { atomicAdd(s_dg + cc + r, tg); atomicAdd(s_db + cc + r, tb); }
                }
            }
        }
    }
    if (LNB) {
        __syncthreads();
        for (int i = tid; i < C; i += MLP_THREADS) { atomicAdd(p.ln_dw + i, s_dg[i]); atomicAdd(p.ln_db + i, s_db[i]); }
    }
}

#define MLP_FWD_RES_DEFAULT 12
static int mlp_cu_count() { return mmg_cu_count_cached(); }

// The backward's 4C-wide outputs (g, dh) beyond the 256 MiB Infinity Cache are consumed much later by the weight-gradient GEMMs:
// stored past L2 (nontemporal) from C = 128 up.  Same-run A/B (profiles/r02_mlp_nt_store_ab.txt): backward -8...-17 % at C = 192,
// -12 % at 384, no change at 96; the forward's saved pre-activation is 1-3 % SLOWER that way and keeps plain stores.
// MMG_MLP_NT_STORE=0/1 forces it off/on everywhere (read per call: A/B runs).
static int mlp_nt_store(long M, int C, bool backward) {
    const char* e = getenv("MMG_MLP_NT_STORE");
    if (e) return atoi(e) != 0;
    return backward && C >= 128 && (size_t)M * 4 * C * 2 >= ((size_t)256 << 20);
}

template <int C, bool RECOMP>
static int launch_mlp_bwd(MlpBwd p, hipStream_t stream) {
    typedef MlpCfg<C> Cfg;
    p.nt = mlp_nt_store(p.M, C, true);
    const size_t lds = 2 * ((RECOMP ? 3 : 2) * Cfg::NC * C * 2) + (size_t)8 * C * sizeof(float);
    p.ntiles = (int)((p.M + Cfg::BM - 1) / Cfg::BM);
    const int cap = Cfg::WGS_BWD * mlp_cu_count();
    const int grid = p.ntiles < cap ? p.ntiles : cap;
    MMG_NOTE_KERNEL("cnblock_mlp_bwd_kernel<%d, %s, %s>", C, RECOMP ? "true" : "false", p.ln_dw ? "true" : "false");
    if (p.ln_dw) {
        mmg_allow_lds(cnblock_mlp_bwd_kernel<C, RECOMP, true>, lds);
        hipLaunchKernelGGL((cnblock_mlp_bwd_kernel<C, RECOMP, true>), dim3(grid), dim3(Cfg::THREADS), lds, stream, p);
    } else {
        mmg_allow_lds(cnblock_mlp_bwd_kernel<C, RECOMP, false>, lds);
        hipLaunchKernelGGL((cnblock_mlp_bwd_kernel<C, RECOMP, false>), dim3(grid), dim3(Cfg::THREADS), lds, stream, p);
    }
    MMG_LAUNCH_CHECK("mmg_cnblock_mlp_bwd");
    return 0;
}

template <int C>
static int launch_mlp_fwd(MlpFwd p, hipStream_t stream) {
    typedef MlpCfg<C> Cfg;
    p.nt = p.hpre ? mlp_nt_store(p.M, C, false) : 0;
    const size_t lds = 2 * (2 * Cfg::NC * C * 2) + (size_t)8 * C * sizeof(float);
    p.ntiles = (int)((p.M + Cfg::BM - 1) / Cfg::BM);
    const int cap = Cfg::WGS * mlp_cu_count();
    const int grid = p.ntiles < cap ? p.ntiles : cap;
    if constexpr (C == 96) {
        // resident weights (see the kernel): MMG_MLP_FWD_RES = waves per workgroup (8 / 12; 0 = the streaming form), read per call
        const char* e = getenv("MMG_MLP_FWD_RES");
        const int rw = e ? atoi(e) : MLP_FWD_RES_DEFAULT;
        if (!p.hpre && (rw == 8 || rw == 12)) {
            const size_t ldsr = (size_t)(4 * C / Cfg::NC) * (2 * Cfg::NC * C * 2) + (size_t)8 * C * sizeof(float);
            p.ntiles = (int)((p.M + 16 * Cfg::MT - 1) / (16 * Cfg::MT));          // 32-row wave tiles
            const int cus = mlp_cu_count();
            const int gridr = (p.ntiles + rw - 1) / rw < cus ? (p.ntiles + rw - 1) / rw : cus;
            MMG_NOTE_KERNEL("cnblock_mlp_fwd_kernel<%d, false, %d>", C, rw);
            if (rw == 12) {
                mmg_allow_lds(cnblock_mlp_fwd_kernel<C, false, 12>, ldsr);
                hipLaunchKernelGGL((cnblock_mlp_fwd_kernel<C, false, 12>), dim3(gridr), dim3(768), ldsr, stream, p);
            } else {
                mmg_allow_lds(cnblock_mlp_fwd_kernel<C, false, 8>, ldsr);
                hipLaunchKernelGGL((cnblock_mlp_fwd_kernel<C, false, 8>), dim3(gridr), dim3(512), ldsr, stream, p);
            }
            MMG_LAUNCH_CHECK("mmg_cnblock_mlp_fwd");
            return 0;
        }
    }
    MMG_NOTE_KERNEL("cnblock_mlp_fwd_kernel<%d, %s>", C, p.hpre ? "true" : "false");
    if (p.hpre) {
        mmg_allow_lds(cnblock_mlp_fwd_kernel<C, true>, lds);
        hipLaunchKernelGGL((cnblock_mlp_fwd_kernel<C, true>), dim3(grid), dim3(Cfg::THREADS), lds, stream, p);
    } else {
        mmg_allow_lds(cnblock_mlp_fwd_kernel<C, false>, lds);
        hipLaunchKernelGGL((cnblock_mlp_fwd_kernel<C, false>), dim3(grid), dim3(Cfg::THREADS), lds, stream, p);
    }
    MMG_LAUNCH_CHECK("mmg_cnblock_mlp_fwd");
    return 0;
}

static bool mlp_supported(int C) { return C == 96 || C == 128 || C == 192 || C == 256 || C == 384 || C == 512; }

MMG_API long long mmg_cnblock_packed_elems(int C, int backward) {
    if (!mlp_supported(C)) return 0;
    return (long long)(backward == 1 ? 3 : 2) * 4 * C * C;
}

MMG_API int mmg_cnblock_pack_weights(const float* w1, const float* w2, const float* gamma, void* packed, int C, int backward,
                                     hipStream_t stream) {
    MMG_CHECK_ARG(w1 && w2 && packed, "mmg_cnblock_pack_weights: null pointer");
    MMG_CHECK_ARG(mlp_supported(C), "mmg_cnblock_pack_weights: C=%d not in {96,128,192,256,384,512}", C);
    MMG_CHECK_ARG(backward >= 0 && backward <= 2, "mmg_cnblock_pack_weights: backward=%d not in {0,1,2}", backward);
    MMG_CHECK_ARG(!backward || gamma, "mmg_cnblock_pack_weights: the backward image needs the layer scale");
    PackArgs a{};
    a.out = (bf16_t*)packed; a.C = C; a.NC = C <= 128 ? 64 : 32;      // = MlpCfg<C>::NC
    a.part[0] = PackPart{w1, 0, nullptr, 0};                       // W1 rows n, K = c            (hidden = x W1^T)
    if (!backward) {
        a.parts = 2;
        a.part[1] = PackPart{w2, 1, nullptr, 1};                   // W2 rows c, K = n            (y = g W2^T)
    } else if (backward == 1) {
        a.parts = 3;
        a.part[1] = PackPart{w2, 1, gamma, 0};                     // gamma*W2^T rows n, K = c    (dG = dy (gamma W2))
        a.part[2] = PackPart{w1, 0, nullptr, 1};                   // W1^T rows c, K = n          (dx = dH W1)
    } else {                                                       // backward == 2: pre-activation read back, no W1 part
        a.parts = 2;
        a.part[0] = PackPart{w2, 1, gamma, 0};
        a.part[1] = PackPart{w1, 0, nullptr, 1};
    }
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(256), dim3(256), 0, stream, a);
    MMG_LAUNCH_CHECK("mmg_cnblock_pack_weights");
    return 0;
}

MMG_API int mmg_cnblock_mlp_fwd(const void* xd, const float* ln_w, const float* ln_b, float eps, const void* packed,
                                const float* b1, const float* b2, const float* gamma, const void* residual, void* y,
                                void* hpre, void* xln, void* gact, float* mean, float* rstd, int hpre_kind, long long M, int C, hipStream_t stream) {
    MMG_CHECK_ARG(xd && ln_w && ln_b && packed && b1 && b2 && gamma && residual && y, "mmg_cnblock_mlp_fwd: null pointer");
    MMG_CHECK_ARG(mlp_supported(C), "mmg_cnblock_mlp_fwd: C=%d not in {96,128,192,256,384,512}", C);
    MMG_CHECK_ARG(M > 0 && M < (1LL << 36), "mmg_cnblock_mlp_fwd: bad M=%lld", M);
    MMG_CHECK_ARG((mean == nullptr) == (hpre == nullptr) && (rstd == nullptr) == (hpre == nullptr),
                  "mmg_cnblock_mlp_fwd: hpre, mean and rstd are saved together or not at all");
    MMG_CHECK_ARG(!xln || hpre, "mmg_cnblock_mlp_fwd: xln is saved next to hpre only");
    MMG_CHECK_ARG(!gact || hpre, "mmg_cnblock_mlp_fwd: gact is saved next to hpre only");
    MlpFwd p{(const bf16_t*)xd, ln_w, ln_b, eps, (const bf16_t*)packed, b1, b2, gamma, (const bf16_t*)residual, (bf16_t*)y,
             (bf16_t*)hpre, (bf16_t*)xln, mean, rstd, (long)M, 0, 0, (bf16_t*)gact, hpre_kind};
    MMG_CHECK_ARG(hpre_kind == 0 || (hpre_kind == 1 && hpre && gact), "mmg_cnblock_mlp_fwd: hpre_kind=%d (1 = GELU' in place of the pre-activation) needs hpre and gact", hpre_kind);
    switch (C) {
        case 96: return launch_mlp_fwd<96>(p, stream);
        case 128: return launch_mlp_fwd<128>(p, stream);
        case 192: return launch_mlp_fwd<192>(p, stream);
        case 256: return launch_mlp_fwd<256>(p, stream);
        case 384: return launch_mlp_fwd<384>(p, stream);
        default: return launch_mlp_fwd<512>(p, stream);
    }
}

// 1: hidden row recomputed (forward saves nothing 4C-wide); 2: needs the forward's saved pre-activation `hpre`; 0: unsupported
MMG_API int mmg_cnblock_mlp_bwd_supported(int C) { return (C == 96 || C == 128 || C == 192) ? 1 : (C == 384 ? 2 : 0); }

MMG_API int mmg_cnblock_mlp_bwd(const void* dy, const void* xd, const float* ln_w, const float* ln_b, float eps,
                                const void* packed_bwd, const float* b1, const void* hpre, void* dh, void* g, void* xln,
                                void* dxln, float* mean, float* rstd, float* ln_dw, float* ln_db, long long M, int C,
                                hipStream_t stream) {
    MMG_CHECK_ARG(dy && xd && ln_w && ln_b && packed_bwd && b1 && dh && g && xln && dxln && mean && rstd,
                  "mmg_cnblock_mlp_bwd: null pointer");
    const int mode = mmg_cnblock_mlp_bwd_supported(C);
    MMG_CHECK_ARG(mode != 0, "mmg_cnblock_mlp_bwd: C=%d not in {96,128,192,384}", C);
    MMG_CHECK_ARG((mode == 2) == (hpre != nullptr), "mmg_cnblock_mlp_bwd: hpre is required for C=384 and unused otherwise (C=%d)", C);
    MMG_CHECK_ARG(M > 0 && M < (1LL << 36), "mmg_cnblock_mlp_bwd: bad M=%lld", M);
    MMG_CHECK_ARG((ln_dw == nullptr) == (ln_db == nullptr), "mmg_cnblock_mlp_bwd: ln_dw and ln_db go together");
    MlpBwd p{(const bf16_t*)dy, (const bf16_t*)xd, ln_w, ln_b, eps, (const bf16_t*)packed_bwd, b1, (bf16_t*)dh, (bf16_t*)g,
             (bf16_t*)xln, (bf16_t*)dxln, mean, rstd, (const bf16_t*)hpre, ln_dw, ln_db, (long)M, 0, 0};
    switch (C) {
        case 96: return launch_mlp_bwd<96, true>(p, stream);
        case 128: return launch_mlp_bwd<128, true>(p, stream);
        case 192: return launch_mlp_bwd<192, true>(p, stream);
        default: return launch_mlp_bwd<384, false>(p, stream);
    }
}
