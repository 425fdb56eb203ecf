// Multi-head self-attention for the BERT text tower (gfx950): head_dim 64, key-padding mask, S <= 512 (forward)
// and S <= 256 (backward).  Replaces HF BertSelfAttention (scores = QK^T/8 + mask, softmax, PV) reached from
// mmgclip/networks/encoder.py:156 (`self.model(**x)`), config in notebooks/bert_experimental.ipynb:609-624.
//
// One workgroup per (sequence, head).  K, V (and Q, dO in the backward) live in LDS as ONE swizzled image each that
// serves both row reads (ds_read_b128: the token is the MFMA row, head_dim is reduced) and transposed reads
// (ds_read_b64_tr_b16: the token is reduced).  Scores are computed "swapped" (A = K tile, B = Q tile) so a lane owns
// one query row: softmax statistics are register-local plus two shuffles, and the probability registers are fed
// straight back as the B operand of P.V with the key order the transposed V read delivers (no LDS round trip for P).
// The backward recomputes P from the saved log-sum-exp (no S x S tensor in HBM), pass 1 with a query row per lane
// (dQ), pass 2 with a key per lane (dK, dV); every output row has exactly one writer, so there are no atomics.
#include "common.h"
#include "dropout.h"

#define ATT_D 64
#define ATT_NEG (-1.0e30f)

// byte offset of 16-byte chunk c (0..7) of token row r in a swizzled [rows][64] bf16 image (128-byte rows)
__device__ __forceinline__ int att_off(int r, int c) { return r * 128 + (((((c >> 1) ^ (r >> 1)) & 3) << 1) | (c & 1)) * 16; }

// stage rows [0,S) of a [.,64]-wide head slice (global row stride ld elements) into the image; rows >= S are zero
__device__ __forceinline__ void att_stage(const bf16_t* __restrict__ src, int ld, char* img, int S, int S_pad) {
    for (int idx = threadIdx.x; idx < S_pad * 8; idx += 256) {
        const int r = idx >> 3, c = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < S) v = *reinterpret_cast<const uint4*>(src + (size_t)r * ld + c * 8);
        *reinterpret_cast<uint4*>(img + att_off(r, c)) = v;
    }
}

// row fragment: lane (li, g) gets img[r0 + li][32*ks + 8*g .. +7]
__device__ __forceinline__ bf16x8 att_row_frag(const char* img, int r0, int ks, int lane) {
    const int r = r0 + (lane & 15), c = 4 * ks + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(img + att_off(r, c));
}

// transposed fragment over 32 tokens starting at t0 for the 16 columns dt*16..: lane (i, g) gets
//   element j<4 : img[t0 + 4g + j][dt*16 + i],   element j>=4 : img[t0 + 16 + 4g + (j-4)][dt*16 + i]
__device__ __forceinline__ bf16x8 att_tr_frag(const char* img, int t0, int dt, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
    const int ra = t0 + 4 * g + q4, rb = ra + 16;
    const int c = dt * 2 + (p >> 1);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(img + att_off(ra, c) + (p & 1) * 8));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(img + att_off(rb, c) + (p & 1) * 8));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// The same fragment(s) for the TILED kernels, which keep the next tile's LDS-DMA in flight while they read this one: issued as inline assembly,
// because hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr BUILTIN whenever an LDS-DMA is outstanding (it cannot tell the two
// halves of the double buffer apart; plain ds_read_b128 does not get that wait) - which drained the prefetch in the middle of every tile (round 4,
// found in csrc/gemm_tn_fp8.hip; same finding as gemm_tn_wide.hip in round 2).  The wait for the data is inside the block: the compiler sees
// finished registers.  Two fragments per block (four reads in flight) where a step needs two.
// Top of a tile iteration in the tiled kernels: the tile requested one iteration ago must have LANDED in LDS before anyone reads it.  __syncthreads()
// does not wait for it - a workgroup-scope fence does not drain the vector-memory counter - and until round 4 the only wait for the DMA in these loops
// was the `s_waitcnt vmcnt(0)` hipcc happens to put in front of the ds_read_tr BUILTIN; with the transposed reads issued as inline assembly (below)
// nothing waited at all, and workgroups read tiles that had not arrived (LSE and context off by rounding-size errors, run to run - found by
// tools/att_check.py).  Explicit now: every wave waits for its own DMA pieces, then the barrier.
__device__ __forceinline__ void att_tile_landed() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ unsigned att_lds_addr(const char* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void att_tr_addr(const char* img, int t0, int dt, int lane, unsigned& a0, unsigned& a1) {
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p = lane & 3;
    const int ra = t0 + 4 * g + q4, rb = ra + 16;
    const int c = dt * 2 + (p >> 1);
    a0 = att_lds_addr(img + att_off(ra, c) + (p & 1) * 8);
    a1 = att_lds_addr(img + att_off(rb, c) + (p & 1) * 8);
}
__device__ __forceinline__ bf16x8 att_tr_frag_tiled(const char* img, int t0, int dt, int lane) {
    unsigned a0, a1;
    att_tr_addr(img, t0, dt, lane, a0, a1);
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1) : "memory");
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void att_tr_frag2_tiled(const char* imgA, const char* imgB, int t0, int dt, int lane, bf16x8& fa, bf16x8& fb) {
    unsigned a0, a1, b0, b1;
    att_tr_addr(imgA, t0, dt, lane, a0, a1);
    att_tr_addr(imgB, t0, dt, lane, b0, b1);
    bf16x4 alo, ahi, blo, bhi;
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(alo), "=&v"(ahi), "=&v"(blo), "=&v"(bhi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "memory");
    fa = bf16x8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
    fb = bf16x8{blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
}

// MMG build knob (A/B, tools/build_ab_lib.sh ... -DATT_ASM_TR=1): the tiled kernels' transposed reads as inline assembly (no compiler-inserted
// vmcnt(0) in front of them; LDS latency exposed per fragment instead) or through the builtin (default: measured faster, profiles/r04_attn_tr_ab.txt)
#ifndef ATT_ASM_TR
#define ATT_ASM_TR 0
#endif
#if ATT_ASM_TR
#define ATT_TR1(img, t0, dt, lane) att_tr_frag_tiled(img, t0, dt, lane)
#define ATT_TR2(ia, ib, t0, dt, lane, fa, fb) att_tr_frag2_tiled(ia, ib, t0, dt, lane, fa, fb)
#else
#define ATT_TR1(img, t0, dt, lane) att_tr_frag(img, t0, dt, lane)
#define ATT_TR2(ia, ib, t0, dt, lane, fa, fb) do { fa = att_tr_frag(ia, t0, dt, lane); fb = att_tr_frag(ib, t0, dt, lane); } while (0)
#endif

__device__ __forceinline__ bf16x8 pack_frag(const f32x4 a, const f32x4 b) {
    bf16x8 r;
    r[0] = (short)f2bf(a[0]); r[1] = (short)f2bf(a[1]); r[2] = (short)f2bf(a[2]); r[3] = (short)f2bf(a[3]);
    r[4] = (short)f2bf(b[0]); r[5] = (short)f2bf(b[1]); r[6] = (short)f2bf(b[2]); r[7] = (short)f2bf(b[3]);
    return r;
}

__device__ __forceinline__ float group4_max(float v) {   // across the 4 lane groups (lane>>4) of one column
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

struct AttArgs {
    const bf16_t* qkv; int ld;          // [B*S, 3*Hd] : q | k | v, head h at columns h*64
    const long long* mask;              // [B,S] 1 = attend, 0 = padding (nullable)
    bf16_t* ctx; int ldc;               // [B*S, Hd]
    float* lse;                         // [B, heads, S]
    int S, S_pad, heads, Hd;             // S = row stride of a sequence (padded layout) / longest sequence (packed layout)
    const int* cu;                      // nullable: packed layout, sequence b = rows cu[b] .. cu[b+1] (all attended)
    float scale;
    int nbh, nblk;                      // flash kernels: 1-D grid of nbh (= B * heads) x nblk row blocks
    // backward
    const bf16_t* dctx; int lddc;       // [B*S, Hd]
    bf16_t* dqkv; int lddq;             // [B*S, 3*Hd]
    DropArgs drop;                      // attention-probability dropout (DROP instantiations of the whole-sequence kernels only)
    int bh0;                            // (sequence, head) index of blockIdx.x = 0 in the mask index (micro-batches of one batch)
};

// keep / scale of probability (sequence-head bh, query q, key k): see dropout.h.  bh goes into the KEY (loop-invariant: one extra
// hash per workgroup), the 18-bit (q, k) position is the index - no 32-bit wrap however many sequences a batch has.
__device__ __forceinline__ float att_drop(float v, int bh, int q, int k, const DropArgs d) {
    const unsigned idx = (unsigned)q * 512u + (unsigned)k;
    return mmg_drop_bits(idx, mmg_drop_key_bh(d.key, (unsigned)bh)) >= d.thresh ? v * d.scale : 0.f;
}

// ---------------------------------------------------------------------------------------------
// DROP: HF's attention_probs_dropout - the normalised probabilities are masked and rescaled before the P V product (the row
// log-sum-exp is that of the undropped softmax).
template <int NT, bool DROP>   // NT >= S_pad / 16
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + a.S_pad * 128;
    float* madd = reinterpret_cast<float*>(smem + 2 * a.S_pad * 128);
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int S = a.cu ? a.cu[b + 1] - a.cu[b] : a.S;                        // this sequence's length
    const size_t row0 = a.cu ? (size_t)a.cu[b] : (size_t)b * a.S;            // its first row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const bf16_t* base = a.qkv + row0 * a.ld + h * ATT_D;
    att_stage(base + a.Hd, a.ld, Ks, S, a.S_pad);
    att_stage(base + 2 * a.Hd, a.ld, Vs, S, a.S_pad);
    for (int k = threadIdx.x; k < a.S_pad; k += 256)
        madd[k] = (k < S && (a.cu || !a.mask || a.mask[(size_t)b * a.S + k] != 0)) ? 0.f : ATT_NEG;
    __syncthreads();

    const int ntile = a.S_pad / 16;
    for (int qt = wave; qt < ntile; qt += 4) {
        const int q0 = qt * 16;
        const int qrow = min(q0 + li, S - 1);
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * a.ld + ks * 32 + g * 8);
        f32x4 sc[NT];
        float m = ATT_NEG;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < ntile) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Ks, t * 16, ks, lane), qf[ks], acc, 0, 0, 0);
                const f32x4 mk = *reinterpret_cast<const f32x4*>(madd + t * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[r] = acc[r] * a.scale + mk[r];
                    m = fmaxf(m, acc[r]);
                }
                sc[t] = acc;
            }
        }
        m = group4_max(m);
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < ntile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[t][r] = __expf(sc[t][r] - m);
                    l += sc[t][r];
                }
            }
        }
        l = group4_sum(l);
        const float inv = 1.0f / l;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NT / 2; ++c) {
            if (2 * c < ntile) {
                f32x4 p0 = sc[2 * c] * inv, p1 = sc[2 * c + 1] * inv;
                if constexpr (DROP) {            // element r of tile t: key 16 t + 4 g + r, query q0 + li
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p0[r] = att_drop(p0[r], blockIdx.x + a.bh0, q0 + li, 32 * c + 4 * g + r, a.drop);
                        p1[r] = att_drop(p1[r], blockIdx.x + a.bh0, q0 + li, 32 * c + 16 + 4 * g + r, a.drop);
                    }
                }
                const bf16x8 pf = pack_frag(p0, p1);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_tr_frag(Vs, c * 32, dt, lane), pf, o[dt], 0, 0, 0);
            }
        }
        if (q0 + li < S) {
            bf16_t* dst = a.ctx + (row0 + q0 + li) * a.ldc + h * ATT_D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 v;
                v.x = pack2bf(o[dt][0], o[dt][1]);
                v.y = pack2bf(o[dt][2], o[dt][3]);
                *reinterpret_cast<uint2*>(dst + dt * 16) = v;
            }
            if (g == 0 && a.lse) a.lse[((size_t)b * a.heads + h) * a.S + q0 + li] = m + __logf(l);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward: dqkv <- (dQ | dK | dV) of this head.  S_pad <= 256.
// ---------------------------------------------------------------------------------------------
// DROP: with P~ = mask P / (1 - p) the forward is O = P~ V, so dV = P~^T dO, dP = mask (dO V^T) / (1 - p) and
// dS = P (dP - delta) with delta = rowsum(dO O) unchanged (sum_k P dP = sum_k P~ (dO V^T) = dO . O).
template <bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int img = a.S_pad * 128;
    char* Qs = smem;
    char* Ks = smem + img;
    char* Vs = smem + 2 * img;
    char* Gs = smem + 3 * img;                                       // dO (gradient of the context)
    float* madd = reinterpret_cast<float*>(smem + 4 * img);          // [S_pad]
    float* lses = madd + a.S_pad;                                    // [S_pad]
    float* delta = lses + a.S_pad;                                   // [S_pad] rowsum(dO * O)
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int S = a.cu ? a.cu[b + 1] - a.cu[b] : a.S;                        // this sequence's length
    const size_t row0 = a.cu ? (size_t)a.cu[b] : (size_t)b * a.S;            // its first row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const bf16_t* base = a.qkv + row0 * a.ld + h * ATT_D;
    const bf16_t* gbase = a.dctx + row0 * a.lddc + h * ATT_D;
    const bf16_t* obase = a.ctx + row0 * a.ldc + h * ATT_D;
    att_stage(base, a.ld, Qs, S, a.S_pad);
    att_stage(base + a.Hd, a.ld, Ks, S, a.S_pad);
    att_stage(base + 2 * a.Hd, a.ld, Vs, S, a.S_pad);
    att_stage(gbase, a.lddc, Gs, S, a.S_pad);
    for (int k = threadIdx.x; k < a.S_pad; k += 256) {
        madd[k] = (k < S && (a.cu || !a.mask || a.mask[(size_t)b * a.S + k] != 0)) ? 0.f : ATT_NEG;
        lses[k] = k < S ? a.lse[((size_t)b * a.heads + h) * a.S + k] : 1.0e30f;
    }
    // delta[q] = sum_d dO[q,d] * O[q,d] : 4 lanes per row, 16 columns each
    for (int idx = threadIdx.x; idx < a.S_pad * 4; idx += 256) {
        const int r = idx >> 2, part = idx & 3;
        float s = 0.f;
        if (r < S) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint4 gv = *reinterpret_cast<const uint4*>(gbase + (size_t)r * a.lddc + part * 16 + c * 8);
                const uint4 ov = *reinterpret_cast<const uint4*>(obase + (size_t)r * a.ldc + part * 16 + c * 8);
                const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) s += bf2f_lo(gw[e]) * bf2f_lo(ow[e]) + bf2f_hi(gw[e]) * bf2f_hi(ow[e]);
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (part == 0) delta[r] = s;
    }
    __syncthreads();

    const int ntile = a.S_pad / 16, nchunk = a.S_pad / 32;
    // ---- pass 1: a query row per lane -> dQ -------------------------------------------------
    for (int qt = wave; qt < ntile; qt += 4) {
        const int q0 = qt * 16;
        const bf16x8 qf0 = att_row_frag(Qs, q0, 0, lane), qf1 = att_row_frag(Qs, q0, 1, lane);
        const bf16x8 gf0 = att_row_frag(Gs, q0, 0, lane), gf1 = att_row_frag(Gs, q0, 1, lane);
        const float lq = lses[q0 + li], dq_ = delta[q0 + li];
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunk; ++c) {
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k0 = c * 32 + t * 16;
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Ks, k0, 0, lane), qf0, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Ks, k0, 1, lane), qf1, s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Vs, k0, 0, lane), gf0, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Vs, k0, 1, lane), gf1, dp, 0, 0, 0);
                const f32x4 mk = *reinterpret_cast<const f32x4*>(madd + k0 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(s[r] * a.scale + mk[r] - lq);
                    const float dpr = DROP ? att_drop(dp[r], blockIdx.x + a.bh0, q0 + li, k0 + 4 * g + r, a.drop) : dp[r];
                    ds[t][r] = p * (dpr - dq_) * a.scale;
                }
            }
            const bf16x8 dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_tr_frag(Ks, c * 32, dt, lane), dsf, dq[dt], 0, 0, 0);
        }
        if (q0 + li < S) {
            bf16_t* dst = a.dqkv + (row0 + q0 + li) * a.lddq + h * ATT_D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 v;
                v.x = pack2bf(dq[dt][0], dq[dt][1]);
                v.y = pack2bf(dq[dt][2], dq[dt][3]);
                *reinterpret_cast<uint2*>(dst + dt * 16) = v;
            }
        }
    }
    // ---- pass 2: a key per lane -> dK, dV ---------------------------------------------------
    for (int kt = wave; kt < ntile; kt += 4) {
        const int k0 = kt * 16;
        const bf16x8 kf0 = att_row_frag(Ks, k0, 0, lane), kf1 = att_row_frag(Ks, k0, 1, lane);
        const bf16x8 vf0 = att_row_frag(Vs, k0, 0, lane), vf1 = att_row_frag(Vs, k0, 1, lane);
        const float mk = madd[k0 + li];
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int c = 0; c < nchunk; ++c) {
            f32x4 pp[2], ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int q0 = c * 32 + t * 16;
                // D[row = q (4g + r)][col = key li]
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Qs, q0, 0, lane), kf0, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Qs, q0, 1, lane), kf1, s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Gs, q0, 0, lane), vf0, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_row_frag(Gs, q0, 1, lane), vf1, dp, 0, 0, 0);
                const f32x4 lq = *reinterpret_cast<const f32x4*>(lses + q0 + 4 * g);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(delta + q0 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(s[r] * a.scale + mk - lq[r]);
                    pp[t][r] = DROP ? att_drop(p, blockIdx.x + a.bh0, q0 + 4 * g + r, k0 + li, a.drop) : p;
                    const float dpr = DROP ? att_drop(dp[r], blockIdx.x + a.bh0, q0 + 4 * g + r, k0 + li, a.drop) : dp[r];
                    ds[t][r] = p * (dpr - dl[r]) * a.scale;
                }
            }
            const bf16x8 pf = pack_frag(pp[0], pp[1]), dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_tr_frag(Gs, c * 32, dt, lane), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_tr_frag(Qs, c * 32, dt, lane), dsf, dk[dt], 0, 0, 0);
            }
        }
        if (k0 + li < S) {
            bf16_t* dst = a.dqkv + (row0 + k0 + li) * a.lddq + h * ATT_D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 v;
                v.x = pack2bf(dk[dt][0], dk[dt][1]);
                v.y = pack2bf(dk[dt][2], dk[dt][3]);
                *reinterpret_cast<uint2*>(dst + a.Hd + dt * 16) = v;
                v.x = pack2bf(dv[dt][0], dv[dt][1]);
                v.y = pack2bf(dv[dt][2], dv[dt][3]);
                *reinterpret_cast<uint2*>(dst + 2 * a.Hd + dt * 16) = v;
            }
        }
    }
}

static int att_check(const char* who, int B, int S, int heads, int Hd, int ld, int smax) {
    MMG_CHECK_ARG(B > 0 && S > 0 && S <= smax, "%s: S=%d must be in [1,%d]", who, S, smax);
    MMG_CHECK_ARG(heads > 0 && Hd == heads * ATT_D, "%s: hidden=%d must equal heads=%d x 64", who, Hd, heads);
    MMG_CHECK_ARG(ld >= 3 * Hd && ld % 8 == 0, "%s: qkv leading dimension %d", who, ld);
    return 0;
}

// ctx = softmax(Q K^T * scale + key_mask) V per head; lse (nullable) receives the row log-sum-exp for the backward.
MMG_API int mmg_attention_fwd(const void* qkv, int ld, const long long* mask, void* ctx, int ldc, float* lse, int B, int S,
                              int heads, int Hd, float scale, hipStream_t stream) {
    if (att_check("mmg_attention_fwd", B, S, heads, Hd, ld, 512)) return 1;
    MMG_CHECK_ARG(qkv && ctx && ldc >= Hd && ldc % 8 == 0, "mmg_attention_fwd: bad ctx");
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.mask = mask; a.ctx = (bf16_t*)ctx; a.ldc = ldc; a.lse = lse;
    a.S = S; a.S_pad = cdiv(S, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    const size_t shm = (size_t)2 * a.S_pad * 128 + a.S_pad * 4;
    const dim3 grid(B * heads);
    const int nt = a.S_pad / 16;
#define ATT_FWD(NT)                                                                                   \
    do {                                                                                              \
        mmg_allow_lds(attn_fwd_kernel<NT, false>, shm);                                               \
        hipLaunchKernelGGL((attn_fwd_kernel<NT, false>), grid, dim3(256), shm, stream, a);            \
    } while (0)
    if (nt <= 6) ATT_FWD(6);
    else if (nt <= 8) ATT_FWD(8);
    else if (nt <= 16) ATT_FWD(16);
    else ATT_FWD(32);
#undef ATT_FWD
    MMG_LAUNCH_CHECK("mmg_attention_fwd");
    return 0;
}

// dqkv[:, q|k|v of every head] = gradients given dctx; needs the forward's ctx and lse.  S <= 256.
MMG_API int mmg_attention_bwd(const void* qkv, int ld, const long long* mask, const void* ctx, int ldc, const float* lse,
                              const void* dctx, int lddc, void* dqkv, int lddq, int B, int S, int heads, int Hd, float scale,
                              hipStream_t stream) {
    if (att_check("mmg_attention_bwd", B, S, heads, Hd, ld, 256)) return 1;
    MMG_CHECK_ARG(qkv && ctx && lse && dctx && dqkv && ldc >= Hd && lddc >= Hd && lddq >= 3 * Hd && ldc % 8 == 0 &&
                      lddc % 8 == 0 && lddq % 8 == 0, "mmg_attention_bwd: bad pointer or leading dimension");
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.mask = mask; a.ctx = (bf16_t*)ctx; a.ldc = ldc; a.lse = const_cast<float*>(lse);
    a.S = S; a.S_pad = cdiv(S, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    a.dctx = (const bf16_t*)dctx; a.lddc = lddc; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq;
    const size_t shm = (size_t)4 * a.S_pad * 128 + 3 * a.S_pad * 4;
    mmg_allow_lds(attn_bwd_kernel<false>, shm);
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(B * heads), dim3(256), shm, stream, a);
    MMG_LAUNCH_CHECK("mmg_attention_bwd");
    return 0;
}

// Packed ("unpadded") layout: the B sequences are stored back to back, sequence b in rows cu_seqlens[b] .. cu_seqlens[b+1]
// (int32 [B+1] on the device), every token attended; S_max bounds the lengths (LDS sizing, lse stride [B, heads, S_max]).
// The reference pads every prompt to max_length (mmgclip/dataset/dataset.py:347) and only reads the [SEP] position
// (mmgclip_model.py:110-111), so the padded rows are pure overhead for the tower.
MMG_API int mmg_attention_varlen_fwd(const void* qkv, int ld, const int* cu_seqlens, void* ctx, int ldc, float* lse, int B,
                                     int S_max, int heads, int Hd, float scale, hipStream_t stream) {
    if (att_check("mmg_attention_varlen_fwd", B, S_max, heads, Hd, ld, 512)) return 1;
    MMG_CHECK_ARG(qkv && ctx && cu_seqlens && ldc >= Hd && ldc % 8 == 0, "mmg_attention_varlen_fwd: bad pointer or ldc");
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.cu = cu_seqlens; a.ctx = (bf16_t*)ctx; a.ldc = ldc; a.lse = lse;
    a.S = S_max; a.S_pad = cdiv(S_max, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    const size_t shm = (size_t)2 * a.S_pad * 128 + a.S_pad * 4;
    const dim3 grid(B * heads);
    const int nt = a.S_pad / 16;
#define ATT_FWD(NT)                                                                                   \
    do {                                                                                              \
        mmg_allow_lds(attn_fwd_kernel<NT, false>, shm);                                               \
        hipLaunchKernelGGL((attn_fwd_kernel<NT, false>), grid, dim3(256), shm, stream, a);            \
    } while (0)
    if (nt <= 6) ATT_FWD(6);
    else if (nt <= 8) ATT_FWD(8);
    else if (nt <= 16) ATT_FWD(16);
    else ATT_FWD(32);
#undef ATT_FWD
    MMG_LAUNCH_CHECK("mmg_attention_varlen_fwd");
    return 0;
}

MMG_API int mmg_attention_varlen_bwd(const void* qkv, int ld, const int* cu_seqlens, const void* ctx, int ldc, const float* lse,
                                     const void* dctx, int lddc, void* dqkv, int lddq, int B, int S_max, int heads, int Hd,
                                     float scale, hipStream_t stream) {
    if (att_check("mmg_attention_varlen_bwd", B, S_max, heads, Hd, ld, 256)) return 1;
    MMG_CHECK_ARG(qkv && ctx && lse && dctx && dqkv && cu_seqlens && ldc >= Hd && lddc >= Hd && lddq >= 3 * Hd && ldc % 8 == 0 &&
                      lddc % 8 == 0 && lddq % 8 == 0, "mmg_attention_varlen_bwd: bad pointer or leading dimension");
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.cu = cu_seqlens; a.ctx = (bf16_t*)ctx; a.ldc = ldc; a.lse = const_cast<float*>(lse);
    a.S = S_max; a.S_pad = cdiv(S_max, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    a.dctx = (const bf16_t*)dctx; a.lddc = lddc; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq;
    const size_t shm = (size_t)4 * a.S_pad * 128 + 3 * a.S_pad * 4;
    mmg_allow_lds(attn_bwd_kernel<false>, shm);
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(B * heads), dim3(256), shm, stream, a);
    MMG_LAUNCH_CHECK("mmg_attention_varlen_bwd");
    return 0;
}

// Training-mode attention with dropout of the probabilities (dropout.h; HF BertSelfAttention.dropout, live in the reference's
// training loop).  One entry point for both layouts: cu_seqlens != nullptr selects the packed layout (mask unused), otherwise the
// padded one with its key mask.  `site` separates the layers' masks; the backward must be given the forward's (p, seed, site).
static int att_drop_args(const char* who, float p, unsigned long long seed, unsigned site, DropArgs& d) {
    MMG_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability %g must be in [0, 1)", who, (double)p);
    d.key = mmg_drop_key(seed, site);
    d.thresh = mmg_drop_threshold(p);
    d.scale = 1.0f / (1.0f - p);
    return 0;
}

MMG_API int mmg_attention_dropout_fwd(const void* qkv, int ld, const long long* mask, const int* cu_seqlens, void* ctx, int ldc,
                                      float* lse, int B, int S, int heads, int Hd, float scale, float p, unsigned long long seed,
                                      unsigned site, int first_sequence, hipStream_t stream) {
    if (att_check("mmg_attention_dropout_fwd", B, S, heads, Hd, ld, 512)) return 1;
    MMG_CHECK_ARG(qkv && ctx && ldc >= Hd && ldc % 8 == 0, "mmg_attention_dropout_fwd: bad ctx");
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.mask = mask; a.cu = cu_seqlens; a.ctx = (bf16_t*)ctx; a.ldc = ldc; a.lse = lse;
    a.S = S; a.S_pad = cdiv(S, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    if (att_drop_args("mmg_attention_dropout_fwd", p, seed, site, a.drop)) return 1;
    MMG_CHECK_ARG(first_sequence >= 0, "mmg_attention_dropout_fwd: first_sequence %d", first_sequence);
    a.bh0 = first_sequence * heads;
    const size_t shm = (size_t)2 * a.S_pad * 128 + a.S_pad * 4;
    const dim3 grid(B * heads);
    const int nt = a.S_pad / 16;
#define ATT_FWD(NT)                                                                                   \
    do {                                                                                              \
        mmg_allow_lds(attn_fwd_kernel<NT, true>, shm);                                                \
        hipLaunchKernelGGL((attn_fwd_kernel<NT, true>), grid, dim3(256), shm, stream, a);             \
    } while (0)
    if (nt <= 6) ATT_FWD(6);
    else if (nt <= 8) ATT_FWD(8);
    else if (nt <= 16) ATT_FWD(16);
    else ATT_FWD(32);
#undef ATT_FWD
    MMG_LAUNCH_CHECK("mmg_attention_dropout_fwd");
    return 0;
}

MMG_API int mmg_attention_dropout_bwd(const void* qkv, int ld, const long long* mask, const int* cu_seqlens, const void* ctx,
                                      int ldc, const float* lse, const void* dctx, int lddc, void* dqkv, int lddq, int B, int S,
                                      int heads, int Hd, float scale, float p, unsigned long long seed, unsigned site,
                                      int first_sequence, hipStream_t stream) {
    if (att_check("mmg_attention_dropout_bwd", B, S, heads, Hd, ld, 256)) return 1;
    MMG_CHECK_ARG(qkv && ctx && lse && dctx && dqkv && ldc >= Hd && lddc >= Hd && lddq >= 3 * Hd && ldc % 8 == 0 &&
                      lddc % 8 == 0 && lddq % 8 == 0, "mmg_attention_dropout_bwd: bad pointer or leading dimension");
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.mask = mask; a.cu = cu_seqlens; a.ctx = (bf16_t*)ctx; a.ldc = ldc;
    a.lse = const_cast<float*>(lse);
    a.S = S; a.S_pad = cdiv(S, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    a.dctx = (const bf16_t*)dctx; a.lddc = lddc; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq;
    if (att_drop_args("mmg_attention_dropout_bwd", p, seed, site, a.drop)) return 1;
    MMG_CHECK_ARG(first_sequence >= 0, "mmg_attention_dropout_bwd: first_sequence %d", first_sequence);
    a.bh0 = first_sequence * heads;
    const size_t shm = (size_t)4 * a.S_pad * 128 + 3 * a.S_pad * 4;
    mmg_allow_lds(attn_bwd_kernel<true>, shm);
    hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(B * heads), dim3(256), shm, stream, a);
    MMG_LAUNCH_CHECK("mmg_attention_dropout_bwd");
    return 0;
}

// =============================================================================================
// Long sequences (ViT-B/16 at 1024x1024: S = 4097; BERT training at S > 256): flash-style tiling.
// Same fragments and LDS images as above, but K/V (forward, dQ) or Q/dO (dK,dV) stream through LDS in 64-row tiles and the
// softmax is computed online; nothing of size S x S exists anywhere.  The backward is two kernels (dQ by query block, dK/dV
// by key block), so every output row has one writer (no atomics, bitwise reproducible).
//   * a wave owns RB blocks of 16 rows: every K / V (Q / dO) fragment read from LDS feeds RB MFMAs (with one block per wave the
//     kernels were LDS-bandwidth bound at 10 % of the MFMA peak: 16 KiB of fragment reads per 16 MFMAs);
//   * tiles arrive by LDS-DMA (global_load_lds, swizzle applied to the source address) into a double buffer: the loads of
//     tile t+1 are in flight while tile t is consumed, one barrier per tile;
//   * softmax in the exp2 domain with the scale folded into one fma per score; the rescale of the running output is skipped
//     while no row maximum of the wave moves;
//   * workgroup -> (sequence, head, block): all blocks of one (sequence, head) run back to back on ONE XCD (XCD = id % 8), so
//     its K / V are fetched from HBM once and served from that XCD's L2 afterwards.
// =============================================================================================
#define ATT_TILE 64
#define ATT_LOG2E 1.4426950408889634f
#define ATT_LN2 0.6931471805599453f

__device__ __forceinline__ void att_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 64 rows x 128 bytes of a head slice -> the swizzled image of att_off() by LDS-DMA.  The destination of a wave's DMA is
// lane-linear (physical chunk p = 8 r + s of the tile), so the thread that fills slot s of row r fetches the logical chunk the
// swizzle (an involution) keeps there.  Rows past the sequence are clamped to its last row: finite values that the callers
// multiply by exactly-zero probabilities.  256 threads, two 16-byte pieces each.
__device__ __forceinline__ void att_stage_dma(const bf16_t* __restrict__ src, int ld, char* img, int row0, int S) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int p = it * 256 + tid, r = p >> 3, sl = p & 7;
        const int c = ((((sl >> 1) ^ (r >> 1)) & 3) << 1) | (sl & 1);
        const int gr = min(row0 + r, S - 1);
        att_glds16(src + (size_t)gr * ld + c * 8, img + (size_t)(it * 256 + (tid & ~63)) * 16);
    }
}

// row fragment straight from global memory: lane (li, g) gets row (r0 + li, clamped)[32*ks + 8*g .. +7]
__device__ __forceinline__ bf16x8 att_global_frag(const bf16_t* __restrict__ base, int ld, int r0, int ks, int lane, int S) {
    const int r = min(r0 + (lane & 15), S - 1);
    return *reinterpret_cast<const bf16x8*>(base + (size_t)r * ld + ks * 32 + (lane >> 4) * 8);
}

// 1-D grid of nbh * nblk workgroups -> ((sequence, head), block)
__device__ __forceinline__ void att_block_map(int nbh, int nblk, int& bh, int& blk) {
    const int L = blockIdx.x;
    if ((nbh & 7) == 0) {
        const int slot = L >> 3;
        bh = (slot / nblk) * 8 + (L & 7);
        blk = slot % nblk;
    } else {
        bh = L / nblk;
        blk = L % nblk;
    }
}

__device__ __forceinline__ float att_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int RB, bool MASK>
__global__ __launch_bounds__(256, 2) void attn_flash_fwd_kernel(const AttArgs a) {
    __shared__ __attribute__((aligned(16))) char KV[2][2][ATT_TILE * 128];
    int bh, qb;
    att_block_map(a.nbh, a.nblk, bh, qb);
    const int b = bh / a.heads, h = bh % a.heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const bf16_t* base = a.qkv + (size_t)b * a.S * a.ld + h * ATT_D;
    const int q0 = qb * (64 * RB) + wave * (16 * RB);
    bf16x8 qf[RB][2];
    float m[RB], l[RB];
    f32x4 o[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        qf[rb][0] = att_global_frag(base, a.ld, q0 + rb * 16, 0, lane, a.S);
        qf[rb][1] = att_global_frag(base, a.ld, q0 + rb * 16, 1, lane, a.S);
        m[rb] = ATT_NEG; l[rb] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[rb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float c2 = a.scale * ATT_LOG2E;
    const int ntiles = (a.S + ATT_TILE - 1) / ATT_TILE;
    att_stage_dma(base + a.Hd, a.ld, KV[0][0], 0, a.S);
    att_stage_dma(base + 2 * a.Hd, a.ld, KV[0][1], 0, a.S);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * ATT_TILE;
        att_tile_landed();     // tile kt has landed (explicit vmcnt(0) of every wave) and nobody still reads the other buffer
        if (kt + 1 < ntiles) {
            att_stage_dma(base + a.Hd, a.ld, KV[(kt + 1) & 1][0], k0 + ATT_TILE, a.S);
            att_stage_dma(base + 2 * a.Hd, a.ld, KV[(kt + 1) & 1][1], k0 + ATT_TILE, a.S);
        }
        const char* Ks = KV[kt & 1][0];
        const char* Vs = KV[kt & 1][1];
        f32x4 sc[RB][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 k0f = att_row_frag(Ks, t * 16, 0, lane), k1f = att_row_frag(Ks, t * 16, 1, lane);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0f, qf[rb][0], acc, 0, 0, 0);
                sc[rb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f, qf[rb][1], acc, 0, 0, 0);
            }
        }
        if (MASK || k0 + ATT_TILE > a.S) {      // (selects, no short-circuit: every mask load is issued before the first use)
            bool ok[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + t * 16 + 4 * g + r;
                    ok[t][r] = key < a.S;
                    if (MASK) ok[t][r] = ok[t][r] & (a.mask[(size_t)b * a.S + min(key, a.S - 1)] != 0);
                }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) sc[rb][t][r] = ok[t][r] ? sc[rb][t][r] : ATT_NEG;
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float tmax = ATT_NEG;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sc[rb][t][r]);
            const float mn = fmaxf(m[rb], group4_max(tmax) * c2);
            const float alpha = att_exp2(m[rb] - mn);
            m[rb] = mn;
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = att_exp2(fmaf(sc[rb][t][r], c2, -mn));
                    sc[rb][t][r] = pv;
                    ps += pv;
                }
            l[rb] = l[rb] * alpha + ps;     // partial over this lane group's keys; groups are summed at the end
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[rb][dt] *= alpha;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 pf[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) pf[rb] = pack_frag(sc[rb][2 * c], sc[rb][2 * c + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vt = ATT_TR1(Vs, c * 32, dt, lane);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) o[rb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pf[rb], o[rb][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float lt = group4_sum(l[rb]);
        const float inv = 1.0f / lt;
        const int q = q0 + rb * 16 + li;
        if (q < a.S) {
            bf16_t* dst = a.ctx + ((size_t)b * a.S + q) * a.ldc + h * ATT_D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 v;
                v.x = pack2bf(o[rb][dt][0] * inv, o[rb][dt][1] * inv);
                v.y = pack2bf(o[rb][dt][2] * inv, o[rb][dt][3] * inv);
                *reinterpret_cast<uint2*>(dst + dt * 16) = v;
            }
            if (g == 0 && a.lse) a.lse[((size_t)b * a.heads + h) * a.S + q] = (m[rb] + __log2f(lt)) * ATT_LN2;
        }
    }
}

// delta[b,h,q] = sum_d dO[q,d] * O[q,d]: 8 lanes per (row, head), 16 bytes per lane (the first version read 2 bytes per lane,
// one wave per pair: 1.7 TB/s)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ ctx, int ldc, const bf16_t* __restrict__ dctx,
                                                         int lddc, float* __restrict__ delta, int B, int S, int heads) {
    const int sub = threadIdx.x & 7;
    const long total = (long)B * S * heads;
    for (long idx = ((long)blockIdx.x * 256 + threadIdx.x) >> 3; idx < total; idx += ((long)gridDim.x * 256) >> 3) {
        const int h = (int)(idx % heads);
        const long row = idx / heads;                 // b*S + q
        float o[8], g[8];
        const uint4 ov = *reinterpret_cast<const uint4*>(ctx + row * ldc + h * ATT_D + sub * 8);
        const uint4 gv = *reinterpret_cast<const uint4*>(dctx + row * lddc + h * ATT_D + sub * 8);
        const unsigned ow[4] = {ov.x, ov.y, ov.z, ov.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[2 * e] = bf2f_lo(ow[e]); o[2 * e + 1] = bf2f_hi(ow[e]); g[2 * e] = bf2f_lo(gw[e]); g[2 * e + 1] = bf2f_hi(gw[e]); }
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v = fmaf(o[e], g[e], v);
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        if (sub == 0) delta[((row / S) * heads + h) * S + (row % S)] = v;
    }
}

// dQ of RB x 64 query rows per workgroup: K / V tiles stream through LDS; p = exp2(s c2 - lse log2e), dS = p (dP - delta) scale
// DROP (BERT training at 256 < S <= 512): dP = mask (dO V^T) / (1 - p), as in attn_bwd_kernel<true>
template <int RB, bool MASK, bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_flash_dq_kernel(const AttArgs a, const float* __restrict__ delta) {
    __shared__ __attribute__((aligned(16))) char KV[2][2][ATT_TILE * 128];
    int bh, qb;
    att_block_map(a.nbh, a.nblk, bh, qb);
    const int b = bh / a.heads, h = bh % a.heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const bf16_t* base = a.qkv + (size_t)b * a.S * a.ld + h * ATT_D;
    const bf16_t* gbase = a.dctx + (size_t)b * a.S * a.lddc + h * ATT_D;
    const int q0 = qb * (64 * RB) + wave * (16 * RB);
    bf16x8 qf[RB][2], gf[RB][2];
    float lq2[RB], dl[RB];
    f32x4 dq[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r0 = q0 + rb * 16;
        qf[rb][0] = att_global_frag(base, a.ld, r0, 0, lane, a.S);
        qf[rb][1] = att_global_frag(base, a.ld, r0, 1, lane, a.S);
        gf[rb][0] = att_global_frag(gbase, a.lddc, r0, 0, lane, a.S);
        gf[rb][1] = att_global_frag(gbase, a.lddc, r0, 1, lane, a.S);
        const bool qok = r0 + li < a.S;
        const size_t sidx = ((size_t)b * a.heads + h) * a.S + min(r0 + li, a.S - 1);
        lq2[rb] = qok ? a.lse[sidx] * ATT_LOG2E : 1.0e30f;
        dl[rb] = qok ? delta[sidx] : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[rb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float c2 = a.scale * ATT_LOG2E;
    const int ntiles = (a.S + ATT_TILE - 1) / ATT_TILE;
    att_stage_dma(base + a.Hd, a.ld, KV[0][0], 0, a.S);
    att_stage_dma(base + 2 * a.Hd, a.ld, KV[0][1], 0, a.S);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * ATT_TILE;
        att_tile_landed();
        if (kt + 1 < ntiles) {
            att_stage_dma(base + a.Hd, a.ld, KV[(kt + 1) & 1][0], k0 + ATT_TILE, a.S);
            att_stage_dma(base + 2 * a.Hd, a.ld, KV[(kt + 1) & 1][1], k0 + ATT_TILE, a.S);
        }
        const char* Ks = KV[kt & 1][0];
        const char* Vs = KV[kt & 1][1];
        const bool edge = MASK || k0 + ATT_TILE > a.S;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 ds[RB][2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int t = 2 * c + t2;
                const bf16x8 k0f = att_row_frag(Ks, t * 16, 0, lane), k1f = att_row_frag(Ks, t * 16, 1, lane);
                const bf16x8 v0f = att_row_frag(Vs, t * 16, 0, lane), v1f = att_row_frag(Vs, t * 16, 1, lane);
                bool ok[4] = {true, true, true, true};
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + t * 16 + 4 * g + r;
                        ok[r] = key < a.S;
                        if (MASK) ok[r] = ok[r] & (a.mask[(size_t)b * a.S + min(key, a.S - 1)] != 0);
                    }
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0f, qf[rb][0], sv, 0, 0, 0);
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f, qf[rb][1], sv, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0f, gf[rb][0], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1f, gf[rb][1], dp, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = att_exp2(fmaf(sv[r], c2, -lq2[rb]));
                        pv = ok[r] ? pv : 0.f;
                        const float dpr = DROP ? att_drop(dp[r], bh + a.bh0, q0 + rb * 16 + li, k0 + t * 16 + 4 * g + r, a.drop) : dp[r];
                        ds[rb][t2][r] = pv * (dpr - dl[rb]) * a.scale;
                    }
                }
            }
            bf16x8 dsf[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) dsf[rb] = pack_frag(ds[rb][0], ds[rb][1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kt_f = ATT_TR1(Ks, c * 32, dt, lane);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) dq[rb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_f, dsf[rb], dq[rb][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int q = q0 + rb * 16 + li;
        if (q < a.S) {
            bf16_t* dst = a.dqkv + ((size_t)b * a.S + q) * a.lddq + h * ATT_D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 v;
                v.x = pack2bf(dq[rb][dt][0], dq[rb][dt][1]);
                v.y = pack2bf(dq[rb][dt][2], dq[rb][dt][3]);
                *reinterpret_cast<uint2*>(dst + dt * 16) = v;
            }
        }
    }
}

// dK, dV of RB x 64 keys per workgroup: Q / dO tiles (+ their lse, delta) stream through LDS
template <int RB, bool MASK, bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_flash_dkv_kernel(const AttArgs a, const float* __restrict__ delta) {
    __shared__ __attribute__((aligned(16))) char QG[2][2][ATT_TILE * 128];
    __shared__ __attribute__((aligned(16))) float lses[2][ATT_TILE];
    __shared__ __attribute__((aligned(16))) float dels[2][ATT_TILE];
    int bh, kb;
    att_block_map(a.nbh, a.nblk, bh, kb);
    const int b = bh / a.heads, h = bh % a.heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const bf16_t* base = a.qkv + (size_t)b * a.S * a.ld + h * ATT_D;
    const bf16_t* gbase = a.dctx + (size_t)b * a.S * a.lddc + h * ATT_D;
    const size_t srow = ((size_t)b * a.heads + h) * a.S;
    const int kbase = kb * (64 * RB) + wave * (16 * RB);
    bf16x8 kf[RB][2], vf[RB][2];
    bool kok[RB];
    f32x4 dk[RB][4], dv[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r0 = kbase + rb * 16;
        kf[rb][0] = att_global_frag(base + a.Hd, a.ld, r0, 0, lane, a.S);
        kf[rb][1] = att_global_frag(base + a.Hd, a.ld, r0, 1, lane, a.S);
        vf[rb][0] = att_global_frag(base + 2 * a.Hd, a.ld, r0, 0, lane, a.S);
        vf[rb][1] = att_global_frag(base + 2 * a.Hd, a.ld, r0, 1, lane, a.S);
        const int key = r0 + li;
        kok[rb] = key < a.S;
        if (MASK) kok[rb] = kok[rb] & (a.mask[(size_t)b * a.S + min(key, a.S - 1)] != 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[rb][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[rb][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    const float c2 = a.scale * ATT_LOG2E;
    const int ntiles = (a.S + ATT_TILE - 1) / ATT_TILE;
    att_stage_dma(base, a.ld, QG[0][0], 0, a.S);
    att_stage_dma(gbase, a.lddc, QG[0][1], 0, a.S);
    if (threadIdx.x < ATT_TILE) {
        const int q = threadIdx.x;
        lses[0][q] = q < a.S ? a.lse[srow + min(q, a.S - 1)] * ATT_LOG2E : 1.0e30f;
        dels[0][q] = q < a.S ? delta[srow + min(q, a.S - 1)] : 0.f;
    }
    for (int qt = 0; qt < ntiles; ++qt) {
        const int qn = (qt + 1) * ATT_TILE;
        att_tile_landed();
        // next tile's row statistics: loaded by every thread from a clamped address and SELECTED where they are stored, at the end of the
        // iteration - a load under `if (thread < 64) if (q < S)` is merged with its default at the end of the branch, and that merge is a use:
        // hipcc waited there with vmcnt(0), i.e. for the LDS-DMA issued just above, in every iteration (round 4)
        // (unconditional too - past the last tile the clamped address re-reads the last row: a value set under `if (qt + 1 < ntiles)` is merged
        //  with its default right after the branch, the same use)
        const size_t qi = srow + min(qn + (int)(threadIdx.x & (ATT_TILE - 1)), a.S - 1);
        const float nl = a.lse[qi], nd = delta[qi];
        if (qt + 1 < ntiles) {
            att_stage_dma(base, a.ld, QG[(qt + 1) & 1][0], qn, a.S);
            att_stage_dma(gbase, a.lddc, QG[(qt + 1) & 1][1], qn, a.S);
        }
        const char* Qs = QG[qt & 1][0];
        const char* Gs = QG[qt & 1][1];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 pp[RB][2], ds[RB][2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int ql = c * 32 + t2 * 16;
                const bf16x8 q0f = att_row_frag(Qs, ql, 0, lane), q1f = att_row_frag(Qs, ql, 1, lane);
                const bf16x8 g0f = att_row_frag(Gs, ql, 0, lane), g1f = att_row_frag(Gs, ql, 1, lane);
                const f32x4 lq = *reinterpret_cast<const f32x4*>(&lses[qt & 1][ql + 4 * g]);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(&dels[qt & 1][ql + 4 * g]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[rb][0], sv, 0, 0, 0);
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[rb][1], sv, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g0f, vf[rb][0], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g1f, vf[rb][1], dp, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = kok[rb] ? att_exp2(fmaf(sv[r], c2, -lq[r])) : 0.f;
                        // (row of this accumulator = query qt * 64 + ql + 4 g + r, column = key kbase + 16 rb + li)
                        pp[rb][t2][r] = DROP ? att_drop(pv, bh + a.bh0, qt * ATT_TILE + ql + 4 * g + r, kbase + rb * 16 + li, a.drop) : pv;
                        const float dpr = DROP ? att_drop(dp[r], bh + a.bh0, qt * ATT_TILE + ql + 4 * g + r, kbase + rb * 16 + li, a.drop) : dp[r];
                        ds[rb][t2][r] = pv * (dpr - dl[r]) * a.scale;
                    }
                }
            }
            bf16x8 pf[RB], dsf[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) { pf[rb] = pack_frag(pp[rb][0], pp[rb][1]); dsf[rb] = pack_frag(ds[rb][0], ds[rb][1]); }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x8 gt, qt_f;
                ATT_TR2(Gs, Qs, c * 32, dt, lane, gt, qt_f);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    dv[rb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf[rb], dv[rb][dt], 0, 0, 0);
                    dk[rb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsf[rb], dk[rb][dt], 0, 0, 0);
                }
            }
        }
        if (qt + 1 < ntiles && threadIdx.x < ATT_TILE) {     // buffer (qt+1)&1 was last read in iteration qt-1
            const bool in = qn + (int)threadIdx.x < a.S;
            lses[(qt + 1) & 1][threadIdx.x] = in ? nl * ATT_LOG2E : 1.0e30f;
            dels[(qt + 1) & 1][threadIdx.x] = in ? nd : 0.f;
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int key = kbase + rb * 16 + li;
        if (key < a.S) {
            bf16_t* dst = a.dqkv + ((size_t)b * a.S + key) * a.lddq + h * ATT_D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 v;
                v.x = pack2bf(dk[rb][dt][0], dk[rb][dt][1]);
                v.y = pack2bf(dk[rb][dt][2], dk[rb][dt][3]);
                *reinterpret_cast<uint2*>(dst + a.Hd + dt * 16) = v;
                v.x = pack2bf(dv[rb][dt][0], dv[rb][dt][1]);
                v.y = pack2bf(dv[rb][dt][2], dv[rb][dt][3]);
                *reinterpret_cast<uint2*>(dst + 2 * a.Hd + dt * 16) = v;
            }
        }
    }
}

// Rows-per-wave override from the environment (a test / tuning knob), snapped to a value that has a kernel instantiation:
// the launch grid is sized from this number, so it must be the template argument actually dispatched.  0 = no override.
static int att_rb_env(const char* name, int max_rb) {
    const char* e = getenv(name);
    if (!e) return 0;
    const int v = atoi(e);
    if (v <= 0) return 0;                       // unparsable or non-positive: use the built-in choice
    if (v >= max_rb) return max_rb;             // forward / dQ: 1, 2, 4; dK/dV: 1, 2, 3
    return v == 3 ? 2 : v;                      // (3 only exists for dK/dV, where max_rb == 3 catches it above)
}

// Tiled (flash-style) attention for any S: same contract as mmg_attention_fwd.
MMG_API int mmg_attention_long_fwd(const void* qkv, int ld, const long long* mask, void* ctx, int ldc, float* lse, int B, int S,
                                   int heads, int Hd, float scale, hipStream_t stream) {
    if (att_check("mmg_attention_long_fwd", B, S, heads, Hd, ld, 1 << 20)) return 1;
    MMG_CHECK_ARG(qkv && ctx && ldc >= Hd && ldc % 8 == 0 && (long)B * heads <= 0x7fffffffL, "mmg_attention_long_fwd: bad ctx");
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.mask = mask; a.ctx = (bf16_t*)ctx; a.ldc = ldc; a.lse = lse;
    a.S = S; a.S_pad = cdiv(S, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    // 64 rows per wave (4 row blocks) once there are enough query blocks to fill the GPU twice over, else 32 / 16
    const int rb_env = att_rb_env("MMG_ATT_RB", 4);      // (read per call: the tests walk through 1 / 2 / 4)
    const long bhn = (long)B * heads;
    const int rb = rb_env ? rb_env : (bhn * cdiv(S, 256) >= 1024 ? 4 : bhn * cdiv(S, 128) >= 1024 ? 2 : 1);
    a.nbh = (int)bhn; a.nblk = cdiv(S, 64 * rb);
    const dim3 grid((unsigned)(bhn * a.nblk));
#define ATT_FWD(RBV)                                                                                         \
    do {                                                                                                     \
        if (mask) hipLaunchKernelGGL((attn_flash_fwd_kernel<RBV, true>), grid, dim3(256), 0, stream, a);     \
        else hipLaunchKernelGGL((attn_flash_fwd_kernel<RBV, false>), grid, dim3(256), 0, stream, a);         \
    } while (0)
    if (rb == 4) ATT_FWD(4); else if (rb == 2) ATT_FWD(2); else ATT_FWD(1);
#undef ATT_FWD
    MMG_LAUNCH_CHECK("mmg_attention_long_fwd");
    return 0;
}

// Backward for any S.  delta_ws: caller-provided fp32 workspace of B*heads*S elements (receives rowsum(dO*O)).
MMG_API int mmg_attention_long_bwd(const void* qkv, int ld, const long long* mask, const void* ctx, int ldc, const float* lse,
                                   const void* dctx, int lddc, void* dqkv, int lddq, float* delta_ws, int B, int S, int heads,
                                   int Hd, float scale, hipStream_t stream) {
    if (att_check("mmg_attention_long_bwd", B, S, heads, Hd, ld, 1 << 20)) return 1;
    MMG_CHECK_ARG(qkv && ctx && lse && dctx && dqkv && delta_ws && ldc >= Hd && lddc >= Hd && lddq >= 3 * Hd && ldc % 8 == 0 &&
                      lddc % 8 == 0 && lddq % 8 == 0, "mmg_attention_long_bwd: bad pointer or leading dimension");
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.mask = mask; a.ctx = (bf16_t*)ctx; a.ldc = ldc; a.lse = const_cast<float*>(lse);
    a.S = S; a.S_pad = cdiv(S, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    a.dctx = (const bf16_t*)dctx; a.lddc = lddc; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq;
    long rows = (long)B * S * heads;
    int blocks = (int)((rows + 31) / 32 > 8192 ? 8192 : (rows + 31) / 32);          // 32 (row, head) pairs per workgroup
    hipLaunchKernelGGL(attn_delta_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)ctx, ldc, (const bf16_t*)dctx, lddc,
                       delta_ws, B, S, heads);
    const int rbq_env = att_rb_env("MMG_ATT_RB", 4);
    const int rbk_env = getenv("MMG_ATT_RB_DKV") ? att_rb_env("MMG_ATT_RB_DKV", 3) : (rbq_env > 2 ? 2 : rbq_env);
    const long bhn = (long)B * heads;
    const int rbq = rbq_env ? rbq_env : (bhn * cdiv(S, 256) >= 1024 ? 4 : bhn * cdiv(S, 128) >= 1024 ? 2 : 1);
    // dK/dV holds two accumulator sets per row block: 3 blocks per wave is what fits 256 registers (measured -4 % vs 2)
    const int rbk = rbk_env ? rbk_env : (bhn * cdiv(S, 192) >= 1024 ? 3 : bhn * cdiv(S, 128) >= 1024 ? 2 : 1);
    const float* dws = (const float*)delta_ws;
    a.nbh = (int)bhn;
#define ATT_BWD(KERNEL, RBV)                                                                                   \
    do {                                                                                                       \
        a.nblk = cdiv(S, 64 * RBV);                                                                            \
        const dim3 grid((unsigned)(bhn * a.nblk));                                                             \
        if (mask) hipLaunchKernelGGL((KERNEL<RBV, true>), grid, dim3(256), 0, stream, a, dws);                 \
        else hipLaunchKernelGGL((KERNEL<RBV, false>), grid, dim3(256), 0, stream, a, dws);                     \
    } while (0)
    if (rbq == 4) ATT_BWD(attn_flash_dq_kernel, 4); else if (rbq == 2) ATT_BWD(attn_flash_dq_kernel, 2); else ATT_BWD(attn_flash_dq_kernel, 1);
    if (rbk == 3) ATT_BWD(attn_flash_dkv_kernel, 3); else if (rbk == 2) ATT_BWD(attn_flash_dkv_kernel, 2); else ATT_BWD(attn_flash_dkv_kernel, 1);
#undef ATT_BWD
    MMG_LAUNCH_CHECK("mmg_attention_long_bwd");
    return 0;
}

// Backward of mmg_attention_dropout_fwd for 256 < S <= 512 (padded layout): the tiled dQ / dK,dV kernels with the same mask.
// delta_ws: caller-provided fp32 workspace of B * heads * S elements.
MMG_API int mmg_attention_dropout_long_bwd(const void* qkv, int ld, const long long* mask, const void* ctx, int ldc, const float* lse,
                                           const void* dctx, int lddc, void* dqkv, int lddq, float* delta_ws, int B, int S, int heads,
                                           int Hd, float scale, float p, unsigned long long seed, unsigned site, int first_sequence,
                                           hipStream_t stream) {
    if (att_check("mmg_attention_dropout_long_bwd", B, S, heads, Hd, ld, 512)) return 1;
    MMG_CHECK_ARG(qkv && ctx && lse && dctx && dqkv && delta_ws && ldc >= Hd && lddc >= Hd && lddq >= 3 * Hd && ldc % 8 == 0 &&
                      lddc % 8 == 0 && lddq % 8 == 0, "mmg_attention_dropout_long_bwd: bad pointer or leading dimension");
    MMG_CHECK_ARG(first_sequence >= 0, "mmg_attention_dropout_long_bwd: first_sequence %d", first_sequence);
    AttArgs a = {};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.mask = mask; a.ctx = (bf16_t*)ctx; a.ldc = ldc; a.lse = const_cast<float*>(lse);
    a.S = S; a.S_pad = cdiv(S, 32) * 32; a.heads = heads; a.Hd = Hd; a.scale = scale;
    a.dctx = (const bf16_t*)dctx; a.lddc = lddc; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq;
    if (att_drop_args("mmg_attention_dropout_long_bwd", p, seed, site, a.drop)) return 1;
    a.bh0 = first_sequence * heads;
    long rows = (long)B * S * heads;
    int blocks = (int)((rows + 31) / 32 > 8192 ? 8192 : (rows + 31) / 32);
    hipLaunchKernelGGL(attn_delta_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)ctx, ldc, (const bf16_t*)dctx, lddc,
                       delta_ws, B, S, heads);
    const long bhn = (long)B * heads;
    const float* dws = (const float*)delta_ws;
    a.nbh = (int)bhn;
    a.nblk = cdiv(S, 64 * 2);                        // two row blocks per wave in both kernels (S <= 512: 128-row workgroups)
    const dim3 grid((unsigned)(bhn * a.nblk));
    if (mask) {
        hipLaunchKernelGGL((attn_flash_dq_kernel<2, true, true>), grid, dim3(256), 0, stream, a, dws);
        hipLaunchKernelGGL((attn_flash_dkv_kernel<2, true, true>), grid, dim3(256), 0, stream, a, dws);
    } else {
        hipLaunchKernelGGL((attn_flash_dq_kernel<2, false, true>), grid, dim3(256), 0, stream, a, dws);
        hipLaunchKernelGGL((attn_flash_dkv_kernel<2, false, true>), grid, dim3(256), 0, stream, a, dws);
    }
    MMG_LAUNCH_CHECK("mmg_attention_dropout_long_bwd");
    return 0;
}
