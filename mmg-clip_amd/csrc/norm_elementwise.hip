// HBM-bound kernels of the encoder towers (gfx950): LayerNorm fwd/bwd, GELU, casts/transposes of the weights,
// global average pool, stem patchify, fused AdamW.  All move 16 bytes per lane and reduce with wave64 shuffles.
//
// Reference call sites these replace (third-party modules the reference instantiates):
//   torchvision LayerNorm2d / nn.LayerNorm inside ConvNeXt  (mmgclip/networks/encoder.py:53 `features`)
//   HF BertSelfOutput/BertOutput/BertEmbeddings LayerNorm   (mmgclip/networks/encoder.py:156)
//   nn.LayerNorm of MLPProjectionHead                        (mmgclip/networks/projection.py:92,100)
//   AdaptiveAvgPool2d                                        (mmgclip/networks/encoder.py:54)
//   input scaling x*65535, (x-32767.5)/32767.5               (mmgclip/networks/image_features.py:95-99)
//   torch.optim.AdamW                                        (mmgclip/experiments/ClassifierExperiment.py:74,118)
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim C of a bf16 [M,C] matrix.  A row is owned by a group of G lanes
// (G = power of two >= C/8, <= 64), 64/G rows per wave, CH 16-byte chunks per lane.
// `patch` != 0 : the normalised row m = (n,h,w) is written to (read from, in the backward) the 2x2-patchified
// position  row (n, h/2, w/2), columns ((h&1)*2 + (w&1))*C ..  of a [M/4, 4C] matrix, so that the following
// 2x2 stride-2 convolution is a plain GEMM (ConvNeXt downsample layers).
// ---------------------------------------------------------------------------------------------
struct LNArgs {
    const bf16_t* x; int ldx;
    const float* gamma; const float* beta; float eps;
    bf16_t* y; int ldy;
    float* mean; float* rstd;
    int M, C;
    int patch, H, W;     // patchified output (H, W = spatial dims of the INPUT rows)
    // backward
    const bf16_t* dy; int lddy;
    bf16_t* dx; int lddx;
    float* dgamma; float* dbeta;
    const bf16_t* add; int ldadd;    // optional residual-path gradient added to dx (pre-LN transformer blocks)
    int nt;                          // stream the output past L2 (tensor larger than the Infinity Cache)
    int y_fp8;                       // y receives OCP e4m3 bytes (ldy in bytes): operand of the fp8 GEMM path
    int x_f32;                       // x holds fp32 (the text tower keeps its pre-LayerNorm sums in fp32); ldx in elements
    float* yf; int ldyf;             // optional fp32 copy of y (the text tower's residual stream), forward only
    const float* res; int ldres;     // optional fp32 residual: the row normalised is x + res, and x is OVERWRITTEN with that sum
};

// 8 consecutive elements of row m of x (bf16 or fp32 storage) as floats
__device__ __forceinline__ void ln_load_x8(const LNArgs& a, int m, int c, float* v) {
    if (a.x_f32) {
        const float* p = reinterpret_cast<const float*>(a.x) + (size_t)m * a.ldx + c * 8;
        const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    } else {
        const uint4 r = *reinterpret_cast<const uint4*>(a.x + (size_t)m * a.ldx + c * 8);
        v[0] = bf2f_lo(r.x); v[1] = bf2f_hi(r.x); v[2] = bf2f_lo(r.y); v[3] = bf2f_hi(r.y);
        v[4] = bf2f_lo(r.z); v[5] = bf2f_hi(r.z); v[6] = bf2f_lo(r.w); v[7] = bf2f_hi(r.w);
    }
}

// returns (size_t)-1 for pixels of an odd last row / column: a 2x2 stride-2 convolution never reads them
__device__ __forceinline__ size_t ln_out_offset(const LNArgs& a, int m, int ld) {
    if (!a.patch) return (size_t)m * ld;
    const int w = m % a.W, h = (m / a.W) % a.H, n = m / (a.W * a.H);
    if (h >= (a.H & ~1) || w >= (a.W & ~1)) return (size_t)-1;
    const size_t prow = ((size_t)n * (a.H / 2) + (h >> 1)) * (a.W / 2) + (w >> 1);
    return prow * ld + ((h & 1) * 2 + (w & 1)) * a.C;
}

// sum over the G lanes of a group (G = 4 ... 64, groups aligned to G), every lane gets the total.  Inside a DPP row of 16 lanes the
// butterfly runs on DPP modifiers (quad_perm, row_half_mirror, row_mirror: plain VALU operations); hipcc lowers every __shfl_xor to
// ds_bpermute_b32 - an LDS-pipe round trip with an exposed lgkmcnt wait each, 12 of them in a row-sized dependent chain per LayerNorm
// row at G = 64 (round 3: the reason these kernels sat at 0.46 of HBM peak).  Only the 16 / 32 steps still cross rows by ds_bpermute.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if (G >= 2) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    if (G >= 4) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    if (G >= 8) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    if (G >= 16) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
    f[0] = bf2f_lo(v.x); f[1] = bf2f_hi(v.x); f[2] = bf2f_lo(v.y); f[3] = bf2f_hi(v.y);
    f[4] = bf2f_lo(v.z); f[5] = bf2f_hi(v.z); f[6] = bf2f_lo(v.w); f[7] = bf2f_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 o;
    o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
    return o;
}

template <int G, int CH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const LNArgs a) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gr = lane / G;
    const int nchunks = a.C / 8;
    const int rows_per_block = 4 * RPW;
    for (int m = blockIdx.x * rows_per_block + wave * RPW + gr; m < a.M; m += gridDim.x * rows_per_block) {
        float v[CH][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = gl + k * G;
            if (c < nchunks) {
                ln_load_x8(a, m, c, v[k]);
                if (a.res) {             // (fp32 x only) pre-LayerNorm sum of the text tower, kept for the backward in place of x
                    const float* rp = a.res + (size_t)m * a.ldres + c * 8;
                    const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                    v[k][0] += r0.x; v[k][1] += r0.y; v[k][2] += r0.z; v[k][3] += r0.w;
                    v[k][4] += r1.x; v[k][5] += r1.y; v[k][6] += r1.z; v[k][7] += r1.w;
                    float* xp = const_cast<float*>(reinterpret_cast<const float*>(a.x)) + (size_t)m * a.ldx + c * 8;
                    *reinterpret_cast<float4*>(xp) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
                    *reinterpret_cast<float4*>(xp + 4) = make_float4(v[k][4], v[k][5], v[k][6], v[k][7]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[k][e];
            }
        }
        s = group_sum<G>(s);
        const float mu = s / a.C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = gl + k * G;
            if (c < nchunks) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mu; q += d * d; }
            }
        }
        q = group_sum<G>(q);
        const float rs = rsqrtf(q / a.C + a.eps);
        const size_t obase = ln_out_offset(a, m, a.ldy);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = gl + k * G;
            if (c < nchunks) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mu) * rs * a.gamma[c * 8 + e] + a.beta[c * 8 + e];
                if (a.y_fp8)
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(a.y) + obase + c * 8) =
                        make_uint2(pack4_e4m3(o[0], o[1], o[2], o[3]), pack4_e4m3(o[4], o[5], o[6], o[7]));
                else if (obase != (size_t)-1) store16_stream(a.y + obase + c * 8, pack8(o), a.nt);
                if (a.yf) {
                    float* q = a.yf + (size_t)m * a.ldyf + c * 8;
                    *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(q + 4) = make_float4(o[4], o[5], o[6], o[7]);
                }
            }
        }
        if (gl == 0) {
            if (a.mean) a.mean[m] = mu;
            if (a.rstd) a.rstd[m] = rs;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma ; dgamma += sum dy*xhat ; dbeta += sum dy
template <int G, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const LNArgs a) {
    constexpr int RPW = 64 / G;
    extern __shared__ float red[];    // [2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gr = lane / G;
    const int nchunks = a.C / 8;
    const int rows_per_block = 4 * RPW;
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) red[i] = 0.f;
    __syncthreads();
    float dg[CH][8], db[CH][8];
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[k][e] = 0.f; db[k][e] = 0.f; }

    for (int m = blockIdx.x * rows_per_block + wave * RPW + gr; m < a.M; m += gridDim.x * rows_per_block) {
        const float mu = a.mean[m], rs = a.rstd[m];
        const size_t gbase = ln_out_offset(a, m, a.lddy);
        float xh[CH][8], g[CH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = gl + k * G;
            if (c < nchunks) {
                float xv[8], dyv[8];
                ln_load_x8(a, m, c, xv);
                unpack8(*reinterpret_cast<const uint4*>(a.dy + gbase + c * 8), dyv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[k][e] = (xv[e] - mu) * rs;
                    g[k][e] = dyv[e] * a.gamma[c * 8 + e];
                    s1 += g[k][e];
                    s2 += g[k][e] * xh[k][e];
                    dg[k][e] += dyv[e] * xh[k][e];
                    db[k][e] += dyv[e];
                }
            }
        }
        s1 = group_sum<G>(s1) / a.C;
        s2 = group_sum<G>(s2) / a.C;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = gl + k * G;
            if (c < nchunks) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (g[k][e] - s1 - xh[k][e] * s2);
                if (a.add) {
                    float r[8];
                    unpack8(*reinterpret_cast<const uint4*>(a.add + (size_t)m * a.ldadd + c * 8), r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += r[e];
                }
                store16_stream(a.dx + (size_t)m * a.lddx + c * 8, pack8(o), a.nt);
            }
        }
    }
    if (a.dgamma) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = gl + k * G;
            if (c < nchunks) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    atomicAdd(&red[c * 8 + e], dg[k][e]);
                    atomicAdd(&red[a.C + c * 8 + e], db[k][e]);
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < a.C; i += 256) {
            atomicAdd(a.dgamma + i, red[i]);
            atomicAdd(a.dbeta + i, red[a.C + i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The image tower's LayerNorms (round 3): bf16 rows of C = 24 G elements, G lanes x 3 chunks per row, every lane busy, no fp32 / fp8 / residual
// variants and therefore no run-time mode branches around the loads.  What bounds these kernels is bytes in flight (tools/micro/hbm_mix.hip:
// a copy needs ~40 KB of reads in flight per CU for 5.3 TB/s): a wave here has 64 / G rows x 3 chunks x 16 B per operand outstanding - 3 KB
// against the 0.75 - 1 KB of the generic kernels at the widths 96 / 192 / 384 (one chunk per lane, a quarter of the lanes idle) - and the
// backward keeps its operands PACKED between its two passes (8 registers per chunk instead of 16 floats), which is what lets three chunks
// per lane fit four waves per SIMD (the generic backward with CH = 3 needed 158 VGPRs and got slower).
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void layernorm_fwd_plain_kernel(const LNArgs a) {
    constexpr int RPW = 64 / G, CH = 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gr = lane / G;
    float gam[CH][8], bet[CH][8];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = gl + k * G;
        const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + c * 8), g1 = *reinterpret_cast<const float4*>(a.gamma + c * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(a.beta + c * 8), b1 = *reinterpret_cast<const float4*>(a.beta + c * 8 + 4);
        gam[k][0] = g0.x; gam[k][1] = g0.y; gam[k][2] = g0.z; gam[k][3] = g0.w; gam[k][4] = g1.x; gam[k][5] = g1.y; gam[k][6] = g1.z; gam[k][7] = g1.w;
        bet[k][0] = b0.x; bet[k][1] = b0.y; bet[k][2] = b0.z; bet[k][3] = b0.w; bet[k][4] = b1.x; bet[k][5] = b1.y; bet[k][6] = b1.z; bet[k][7] = b1.w;
    }
    const float inv_c = 1.0f / (float)a.C;
    const int rows_per_block = 4 * RPW;
    for (int m = blockIdx.x * rows_per_block + wave * RPW + gr; m < a.M; m += gridDim.x * rows_per_block) {
        uint4 raw[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) raw[k] = *reinterpret_cast<const uint4*>(a.x + (size_t)m * a.ldx + (gl + k * G) * 8);
        float v[CH][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            unpack8(raw[k], v[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[k][e];
        }
        const float mu = group_sum<G>(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[k][e] -= mu; q += v[k][e] * v[k][e]; }
        const float rs = rsqrtf(group_sum<G>(q) * inv_c + a.eps);
        const size_t obase = ln_out_offset(a, m, a.ldy);
        if (obase != (size_t)-1) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = v[k][e] * rs * gam[k][e] + bet[k][e];
                store16_stream(a.y + obase + (gl + k * G) * 8, pack8(o), a.nt);
            }
        }
        if (gl == 0) {
            if (a.mean) a.mean[m] = mu;
            if (a.rstd) a.rstd[m] = rs;
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void layernorm_bwd_plain_kernel(const LNArgs a) {
    constexpr int RPW = 64 / G, CH = 3;
    extern __shared__ float red[];    // [2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gr = lane / G;
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) red[i] = 0.f;
    __syncthreads();
    float gam[CH][8], dg[CH][8], db[CH][8];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int c = gl + k * G;
        const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + c * 8), g1 = *reinterpret_cast<const float4*>(a.gamma + c * 8 + 4);
        gam[k][0] = g0.x; gam[k][1] = g0.y; gam[k][2] = g0.z; gam[k][3] = g0.w; gam[k][4] = g1.x; gam[k][5] = g1.y; gam[k][6] = g1.z; gam[k][7] = g1.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[k][e] = 0.f; db[k][e] = 0.f; }
    }
    const float inv_c = 1.0f / (float)a.C;
    const int rows_per_block = 4 * RPW;
    for (int m = blockIdx.x * rows_per_block + wave * RPW + gr; m < a.M; m += gridDim.x * rows_per_block) {
        const size_t gbase = ln_out_offset(a, m, a.lddy);
        uint4 xr[CH], dr[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            xr[k] = *reinterpret_cast<const uint4*>(a.x + (size_t)m * a.ldx + (gl + k * G) * 8);
            dr[k] = *reinterpret_cast<const uint4*>(a.dy + gbase + (gl + k * G) * 8);
        }
        const float mu = a.mean[m], rs = a.rstd[m];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            float xv[8], dyv[8];
            unpack8(xr[k], xv);
            unpack8(dr[k], dyv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - mu) * rs, g = dyv[e] * gam[k][e];
                s1 += g;
                s2 += g * xh;
                dg[k][e] += dyv[e] * xh;
                db[k][e] += dyv[e];
            }
        }
        s1 = group_sum<G>(s1) * inv_c;
        s2 = group_sum<G>(s2) * inv_c;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            float xv[8], dyv[8], o[8];
            unpack8(xr[k], xv);
            unpack8(dr[k], dyv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rs * (dyv[e] * gam[k][e] - s1 - (xv[e] - mu) * rs * s2);
            store16_stream(a.dx + (size_t)m * a.lddx + (gl + k * G) * 8, pack8(o), a.nt);
        }
    }
    if (a.dgamma) {
#pragma unroll
        for (int k = 0; k < CH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                atomicAdd(&red[(gl + k * G) * 8 + e], dg[k][e]);
                atomicAdd(&red[a.C + (gl + k * G) * 8 + e], db[k][e]);
            }
        __syncthreads();
        for (int i = threadIdx.x; i < a.C; i += 256) {
            atomicAdd(a.dgamma + i, red[i]);
            atomicAdd(a.dbeta + i, red[a.C + i]);
        }
    }
}

template <bool BWD>
static int launch_ln(const LNArgs& a, hipStream_t stream) {
    const int nch = a.C / 8;
    int G = 8;
    while (G < nch && G < 64) G <<= 1;
    int ch = cdiv(nch, G);
    // round 3, FORWARD only: widths of 3 x 2^k chunks (192, 384, 768, 1536: every ConvNeXt-T / BERT width but 96) take G = 2^k lanes x 3
    // chunks per row instead of the next power of two with a quarter of the lanes idle: no idle lanes, 64 / G rows and three times the bytes
    // in flight per wave.  Same-run A/B (tools/ln_ab.py, profiles/r03_ln_ab.txt): forward 498 -> 389 us at M = 1 M x 384 (3.2 -> 4.1 TB/s),
    // 213 -> 191 at 262 k x 768, 779 -> 745 at 4.2 M x 192; the BACKWARD got slower with it (652 -> 805, 1231 -> 1523, 321 -> 432 us: three
    // chunks of dgamma / dbeta accumulators and operands per lane) and keeps the round-2 selection.  MMG_LN_CH3=0: off (A/B).
    static const int ch3_on = getenv("MMG_LN_CH3") ? atoi(getenv("MMG_LN_CH3")) : 1;
    const bool ch3 = !BWD && ch3_on && nch % 3 == 0 && (nch / 3 == 8 || nch / 3 == 16 || nch / 3 == 32 || nch / 3 == 64);
    if (ch3) { G = nch / 3; ch = 3; }
    // the image tower's widths on the plain kernels above (MMG_LN_PLAIN=0: off, A/B)
    static const int plain_on = getenv("MMG_LN_PLAIN") ? atoi(getenv("MMG_LN_PLAIN")) : 1;
    const int g3 = nch % 3 == 0 ? nch / 3 : 0;
    const bool plain = plain_on && (g3 == 4 || g3 == 8 || g3 == 16 || g3 == 32 || g3 == 64) && !a.x_f32 && !a.res && !a.y_fp8 && !a.yf && !a.add &&
                       (!BWD || (a.mean && a.rstd));
    if (plain) G = g3;
    const int rows_per_block = 4 * (64 / G);
    int blocks = cdiv(a.M, rows_per_block);
    const int cap = BWD ? 1024 : 4096;
    if (blocks > cap) blocks = cap;
    const size_t shm = BWD ? 2 * a.C * sizeof(float) : 0;
    if (plain) {
#define LN_PLAIN(GG)                                                                                                      \
    do {                                                                                                                  \
        MMG_NOTE_KERNEL(BWD ? "layernorm_bwd_plain_kernel<%d>" : "layernorm_fwd_plain_kernel<%d>", GG);                   \
        if (BWD) hipLaunchKernelGGL((layernorm_bwd_plain_kernel<GG>), dim3(blocks), dim3(256), shm, stream, a);           \
        else hipLaunchKernelGGL((layernorm_fwd_plain_kernel<GG>), dim3(blocks), dim3(256), 0, stream, a);                 \
    } while (0)
        if (G == 4) LN_PLAIN(4);
        else if (G == 8) LN_PLAIN(8);
        else if (G == 16) LN_PLAIN(16);
        else if (G == 32) LN_PLAIN(32);
        else LN_PLAIN(64);
#undef LN_PLAIN
        return 0;
    }
#define LN_LAUNCH(GG, CC)                                                                                       \
    do {                                                                                                        \
        MMG_NOTE_KERNEL(BWD ? "layernorm_bwd_kernel<%d, %d>" : "layernorm_fwd_kernel<%d, %d>", GG, CC);         \
        if (BWD) hipLaunchKernelGGL((layernorm_bwd_kernel<GG, CC>), dim3(blocks), dim3(256), shm, stream, a);  \
        else hipLaunchKernelGGL((layernorm_fwd_kernel<GG, CC>), dim3(blocks), dim3(256), 0, stream, a);        \
    } while (0)
    if (ch3 && G == 8) LN_LAUNCH(8, 3);
    else if (ch3 && G == 16) LN_LAUNCH(16, 3);
    else if (ch3 && G == 32) LN_LAUNCH(32, 3);
    else if (ch3) LN_LAUNCH(64, 3);
    else if (G == 8) LN_LAUNCH(8, 1);
    else if (G == 16) LN_LAUNCH(16, 1);
    else if (G == 32) LN_LAUNCH(32, 1);
    else if (ch <= 1) LN_LAUNCH(64, 1);
    else if (ch <= 2) LN_LAUNCH(64, 2);
    else if (ch <= 4) LN_LAUNCH(64, 4);
    else if (ch <= 8) LN_LAUNCH(64, 8);
    else { mmg_set_error("layernorm: C=%d too wide (max 4096)", a.C); return 1; }
#undef LN_LAUNCH
    return 0;
}

static int ln_check(const char* who, int M, int C, int patch, int H, int W) {
    MMG_CHECK_ARG(M > 0 && C >= 8 && C % 8 == 0 && C <= 4096, "%s: M=%d C=%d (C must be a multiple of 8, <= 4096)", who, M, C);
    MMG_CHECK_ARG(!patch || (H > 1 && W > 1 && M % (H * W) == 0), "%s: patchified layout needs H=%d W=%d dividing M=%d", who, H, W, M);
    return 0;
}

MMG_API int mmg_layernorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, float eps, void* y, int ldy,
                              float* mean, float* rstd, int M, int C, int patch, int H, int W, hipStream_t stream) {
    if (ln_check("mmg_layernorm_fwd", M, C, patch, H, W)) return 1;
    MMG_CHECK_ARG(x && gamma && beta && y && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= (patch ? 4 * C : C),
                  "mmg_layernorm_fwd: bad pointer or leading dimension");
    LNArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.eps = eps; a.y = (bf16_t*)y; a.ldy = ldy;
    a.mean = mean; a.rstd = rstd; a.M = M; a.C = C; a.patch = patch; a.H = H; a.W = W;
    a.nt = (size_t)M * C * 2 >= ((size_t)256 << 20);
    if (launch_ln<false>(a, stream)) return 1;
    MMG_LAUNCH_CHECK("mmg_layernorm_fwd");
    return 0;
}

// LayerNorm of an fp32 input row (the text tower's pre-LayerNorm sums x + f(x) stay in fp32: bf16 rounding of the residual
// stream is what owned the end-to-end loss error at BASELINE config C1); y bf16 (GEMM operand) + optional fp32 copy yf.
// With `res` the residual add happens HERE (x <- x + res in place, then normalised): the GEMM that produced x writes plain fp32
// and its epilogue stays untouched (a runtime fp32-residual branch there cost the big ConvNeXt GEMMs 4-6 %).
MMG_API int mmg_layernorm_fwd_f32(float* x, int ldx, const float* res, int ldres, const float* gamma, const float* beta, float eps,
                                  void* y, int ldy, float* yf, int ldyf, float* mean, float* rstd, int M, int C, hipStream_t stream) {
    if (ln_check("mmg_layernorm_fwd_f32", M, C, 0, 0, 0)) return 1;
    MMG_CHECK_ARG(x && gamma && beta && y && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C && (!yf || (ldyf >= C && ldyf % 4 == 0)) &&
                      (!res || (ldres >= C && ldres % 4 == 0)), "mmg_layernorm_fwd_f32: bad pointer or leading dimension");
    LNArgs a = {};
    a.x = reinterpret_cast<const bf16_t*>(x); a.x_f32 = 1; a.ldx = ldx; a.res = res; a.ldres = ldres; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.y = (bf16_t*)y; a.ldy = ldy; a.yf = yf; a.ldyf = ldyf; a.mean = mean; a.rstd = rstd; a.M = M; a.C = C;
    if (launch_ln<false>(a, stream)) return 1;
    MMG_LAUNCH_CHECK("mmg_layernorm_fwd_f32");
    return 0;
}
MMG_API int mmg_layernorm_bwd_f32(const void* dy, int lddy, const float* x, int ldx, const float* mean, const float* rstd,
                                  const float* gamma, void* dx, int lddx, float* dgamma, float* dbeta, int M, int C,
                                  const void* add, int ldadd, hipStream_t stream) {
    if (ln_check("mmg_layernorm_bwd_f32", M, C, 0, 0, 0)) return 1;
    MMG_CHECK_ARG(dy && x && mean && rstd && gamma && dx && ((dgamma == nullptr) == (dbeta == nullptr)) && ldx % 8 == 0 &&
                      lddy % 8 == 0 && lddx % 8 == 0 && ldx >= C && lddx >= C && (!add || (ldadd >= C && ldadd % 8 == 0)),
                  "mmg_layernorm_bwd_f32: bad pointer or leading dimension");
    LNArgs a = {};
    a.x = reinterpret_cast<const bf16_t*>(x); a.x_f32 = 1; a.ldx = ldx; a.gamma = gamma;
    a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.M = M; a.C = C;
    a.dy = (const bf16_t*)dy; a.lddy = lddy; a.dx = (bf16_t*)dx; a.lddx = lddx; a.dgamma = dgamma; a.dbeta = dbeta;
    a.add = (const bf16_t*)add; a.ldadd = ldadd;
    if (launch_ln<true>(a, stream)) return 1;
    MMG_LAUNCH_CHECK("mmg_layernorm_bwd_f32");
    return 0;
}

// LayerNorm whose output is written as e4m3 bytes (unscaled, saturating): the A operand of mmg_gemm_nt_fp8
MMG_API int mmg_layernorm_fwd_fp8(const void* x, int ldx, const float* gamma, const float* beta, float eps, void* y8, int ldy,
                                  float* mean, float* rstd, int M, int C, hipStream_t stream) {
    if (ln_check("mmg_layernorm_fwd_fp8", M, C, 0, 0, 0)) return 1;
    MMG_CHECK_ARG(x && gamma && beta && y8 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C,
                  "mmg_layernorm_fwd_fp8: bad pointer or leading dimension");
    LNArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.eps = eps; a.y = (bf16_t*)y8; a.ldy = ldy;
    a.mean = mean; a.rstd = rstd; a.M = M; a.C = C; a.y_fp8 = 1;
    if (launch_ln<false>(a, stream)) return 1;
    MMG_LAUNCH_CHECK("mmg_layernorm_fwd_fp8");
    return 0;
}

MMG_API int mmg_layernorm_bwd(const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                              const float* gamma, void* dx, int lddx, float* dgamma, float* dbeta, int M, int C,
                              int patch, int H, int W, const void* add, int ldadd, hipStream_t stream) {
    if (ln_check("mmg_layernorm_bwd", M, C, patch, H, W)) return 1;
    MMG_CHECK_ARG(!patch || (H % 2 == 0 && W % 2 == 0), "mmg_layernorm_bwd: training through the patchified layout needs even H=%d W=%d", H, W);
    MMG_CHECK_ARG(dy && x && mean && rstd && gamma && dx && ((dgamma == nullptr) == (dbeta == nullptr)) &&
                      ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ldx >= C && lddx >= C,
                  "mmg_layernorm_bwd: bad pointer or leading dimension");
    LNArgs a = {};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.gamma = gamma; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
    a.M = M; a.C = C; a.patch = patch; a.H = H; a.W = W;
    a.dy = (const bf16_t*)dy; a.lddy = lddy; a.dx = (bf16_t*)dx; a.lddx = lddx; a.dgamma = dgamma; a.dbeta = dbeta;
    a.add = (const bf16_t*)add; a.ldadd = ldadd;
    a.nt = (size_t)M * C * 2 >= ((size_t)256 << 20);
    MMG_CHECK_ARG(!add || (ldadd >= C && ldadd % 8 == 0), "mmg_layernorm_bwd: bad ldadd=%d", ldadd);
    if (launch_ln<true>(a, stream)) return 1;
    MMG_LAUNCH_CHECK("mmg_layernorm_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// elementwise
// ---------------------------------------------------------------------------------------------
// y = gelu(x) (bf16 -> bf16), n % 8 == 0; used to rebuild the FFN activation in the backward pass
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        float f[8];
        unpack8(reinterpret_cast<const uint4*>(x)[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = gelu_bf16(f[e]);
        reinterpret_cast<uint4*>(y)[i] = pack8(f);
    }
}

MMG_API int mmg_gelu_fwd_bf16(const void* x, void* y, long long n, hipStream_t stream) {
    MMG_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "mmg_gelu_fwd_bf16: n=%lld must be a positive multiple of 8", n);
    const size_t nvec = (size_t)n / 8;
    int blocks = (int)((nvec + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, nvec);
    MMG_LAUNCH_CHECK("mmg_gelu_fwd_bf16");
    return 0;
}

// fp32 -> bf16 (any n)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = f2bf(x[i]);
}
MMG_API int mmg_cast_f32_bf16(const float* x, void* y, long long n, hipStream_t stream) {
    MMG_CHECK_ARG(x && y && n > 0, "mmg_cast_f32_bf16: bad argument");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, stream, x, (bf16_t*)y, (size_t)n);
    MMG_LAUNCH_CHECK("mmg_cast_f32_bf16");
    return 0;
}
// bf16 -> fp32
__global__ __launch_bounds__(256) void cast_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = bf2f(x[i]);
}
MMG_API int mmg_cast_bf16_f32(const void* x, float* y, long long n, hipStream_t stream) {
    MMG_CHECK_ARG(x && y && n > 0, "mmg_cast_bf16_f32: bad argument");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_f32_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, y, (size_t)n);
    MMG_LAUNCH_CHECK("mmg_cast_bf16_f32");
    return 0;
}

// dst[c, r] (bf16, ld = ldd) = rowscale[r] * src[r, c]  (fp32 [R,C]) : transposed bf16 copy of a weight for the
// data-gradient GEMM; rowscale (nullable) folds ConvNeXt's layer scale into it.
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, int R, int C,
                                                             const float* __restrict__ rowscale,
                                                             bf16_t* __restrict__ dst, int ldd) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < C) ? src[(size_t)r * C + c] * (rowscale ? rowscale[r] : 1.f) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * ldd + r] = f2bf(tile[tx][j]);
    }
}
MMG_API int mmg_transpose_cast_bf16(const float* src, int R, int C, const float* rowscale, void* dst, int ldd,
                                    hipStream_t stream) {
    MMG_CHECK_ARG(src && dst && R > 0 && C > 0 && ldd >= R, "mmg_transpose_cast_bf16: bad argument");
    hipLaunchKernelGGL(transpose_cast_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, stream, src, R, C, rowscale,
                       (bf16_t*)dst, ldd);
    MMG_LAUNCH_CHECK("mmg_transpose_cast_bf16");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// global average pool over the HW rows of each image: x bf16 [n, HW, C] -> y [n, C] (fp32) ; backward broadcast
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int HW, int C) {
    extern __shared__ float sums[];   // [C]
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < C; i += 256) sums[i] = 0.f;
    __syncthreads();
    const int ng = C / 8;
    const int cg = threadIdx.x % ng, rl = threadIdx.x / ng, lanes_r = 256 / ng;
    if (rl < lanes_r) {
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = rl + blockIdx.y * lanes_r; p < HW; p += lanes_r * gridDim.y) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(x + ((size_t)n * HW + p) * C + cg * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&sums[cg * 8 + e], s[e]);
    }
    __syncthreads();
    const float inv = 1.0f / HW;
    for (int i = threadIdx.x; i < C; i += 256) atomicAdd(y + (size_t)n * C + i, sums[i] * inv);
}
MMG_API int mmg_avgpool_fwd(const void* x, float* y, int n, int HW, int C, hipStream_t stream) {
    MMG_CHECK_ARG(x && y && n > 0 && HW > 0 && C % 8 == 0 && C >= 8 && C <= 2048, "mmg_avgpool_fwd: bad argument");
    if (hipMemsetAsync(y, 0, (size_t)n * C * sizeof(float), stream) != hipSuccess) { mmg_set_error("mmg_avgpool_fwd: memset failed"); return 2; }
    int split = cdiv(HW, 256);
    if (split > 16) split = 16;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(n, split), dim3(256), C * sizeof(float), stream, (const bf16_t*)x, y, HW, C);
    MMG_LAUNCH_CHECK("mmg_avgpool_fwd");
    return 0;
}
// dx[n, p, c] = dy[n, c] / HW  (bf16 out)
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, bf16_t* __restrict__ dx, int HW, int C,
                                                          size_t nvec) {
    const int ng = C / 8;
    const float inv = 1.0f / HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % ng);
        const size_t row = i / ng;
        const size_t n = row / HW;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = dy[n * C + cg * 8 + e] * inv;
        reinterpret_cast<uint4*>(dx)[i] = pack8(f);
    }
}
MMG_API int mmg_avgpool_bwd(const float* dy, void* dx, int n, int HW, int C, hipStream_t stream) {
    MMG_CHECK_ARG(dy && dx && n > 0 && HW > 0 && C % 8 == 0, "mmg_avgpool_bwd: bad argument");
    const size_t nvec = (size_t)n * HW * C / 8;
    int blocks = (int)((nvec + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dy, (bf16_t*)dx, HW, C, nvec);
    MMG_LAUNCH_CHECK("mmg_avgpool_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// stem patchify: pixels [n, Cin, H, W] fp32 in [0,1] -> rows (n, h/P, w/P) x (kh, kw, cin) bf16, K padded to Kp.
// `scale16` applies the reference's 16-bit scaling ((65535 x) - 32767.5) / 32767.5 (image_features.py:95-99).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int Cin,
                                                       int H, int W, int P, int Kp, int scale16, size_t rows) {
    const int Ho = H / P, Wo = W / P;
    const int K = P * P * Cin;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < rows * Kp; idx += (size_t)gridDim.x * 256) {
        const int k = (int)(idx % Kp);
        const size_t row = idx / Kp;
        float v = 0.f;
        if (k < K) {
            const int ci = k % Cin, kw = (k / Cin) % P, kh = k / (Cin * P);
            const int wo = (int)(row % Wo), ho = (int)((row / Wo) % Ho);
            const size_t n = row / ((size_t)Wo * Ho);
            v = img[((n * Cin + ci) * H + (ho * P + kh)) * W + wo * P + kw];
            if (scale16) v = (v * 65535.0f - 32767.5f) / 32767.5f;
        }
        out[idx] = f2bf(v);
    }
}
// Cin = 1, P = 4 (the mammography stem): one thread per output row - four 16-byte loads (one per kernel row, consecutive lanes
// = consecutive patches of an image row), 32 bytes of bf16 out + zero padding, all as 16-byte stores.
__global__ __launch_bounds__(256) void patchify_1x4_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int H, int W,
                                                           int Kp, int scale16, size_t rows) {
    const int Ho = H / 4, Wo = W / 4;
    for (size_t row = (size_t)blockIdx.x * 256 + threadIdx.x; row < rows; row += (size_t)gridDim.x * 256) {
        const int wo = (int)(row % Wo), ho = (int)((row / Wo) % Ho);
        const size_t n = row / ((size_t)Wo * Ho);
        const float* src = img + (n * H + (size_t)ho * 4) * W + wo * 4;
        unsigned pk[8];
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
            float4 v = *reinterpret_cast<const float4*>(src + (size_t)kh * W);
            if (scale16) {
                v.x = (v.x * 65535.0f - 32767.5f) / 32767.5f; v.y = (v.y * 65535.0f - 32767.5f) / 32767.5f;
                v.z = (v.z * 65535.0f - 32767.5f) / 32767.5f; v.w = (v.w * 65535.0f - 32767.5f) / 32767.5f;
            }
            pk[2 * kh] = pack2bf(v.x, v.y); pk[2 * kh + 1] = pack2bf(v.z, v.w);
        }
        uint4* dst = reinterpret_cast<uint4*>(out + row * Kp);
        dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        for (int z = 2; z < Kp / 8; ++z) dst[z] = make_uint4(0, 0, 0, 0);
    }
}

MMG_API int mmg_patchify(const float* img, void* out, int n, int Cin, int H, int W, int P, int Kp, int scale16,
                         hipStream_t stream) {
    MMG_CHECK_ARG(img && out && n > 0 && Cin > 0 && P > 0 && H >= P && W >= P && Kp >= P * P * Cin && Kp % 8 == 0,
                  "mmg_patchify: bad argument (H=%d W=%d P=%d Cin=%d Kp=%d)", H, W, P, Cin, Kp);
    const size_t rows = (size_t)n * (H / P) * (W / P);
    if (Cin == 1 && P == 4 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0) {
        const int blocks = (int)((rows + 255) / 256 > 16384 ? 16384 : (rows + 255) / 256);
        hipLaunchKernelGGL(patchify_1x4_kernel, dim3(blocks), dim3(256), 0, stream, img, (bf16_t*)out, H, W, Kp, scale16, rows);
        MMG_LAUNCH_CHECK("mmg_patchify");
        return 0;
    }
    size_t total = rows * Kp;
    int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, stream, img, (bf16_t*)out, Cin, H, W, P, Kp, scale16, rows);
    MMG_LAUNCH_CHECK("mmg_patchify");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// fused AdamW over a flat fp32 parameter buffer (torch.optim.AdamW semantics, decoupled weight decay);
// optionally refreshes the bf16 working copy in the same pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ p16, size_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                                    float gscale) {
    const float step = lr / bc1;
    const float isbc2 = rsqrtf(bc2);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        pi -= step * mi / (sqrtf(vi) * isbc2 + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (p16) p16[i] = f2bf(pi);
    }
}
MMG_API int mmg_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, float grad_scale, hipStream_t stream) {
    MMG_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "mmg_adamw_step: bad argument");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, (size_t)n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, grad_scale);
    MMG_LAUNCH_CHECK("mmg_adamw_step");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// ConvNeXt layer scale: out = x + gamma * (G W2^T + b2).  The weight-gradient GEMM produces the UNSCALED
// dW2raw = dOut^T G and db2raw = colsum(dOut); this kernel turns them into the true gradients (one wave per row c):
//   dgamma[c] += <W2[c,:], dW2raw[c,:]> + b2[c] * db2raw[c] ;  dW2[c,:] += gamma[c] * dW2raw[c,:] ;  db2[c] += gamma[c] * db2raw[c]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layerscale_finalize_kernel(const float* __restrict__ W2, const float* __restrict__ b2,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ dW2raw,
                                                                  const float* __restrict__ db2raw, float* __restrict__ dW2,
                                                                  float* __restrict__ db2, float* __restrict__ dgamma, int C,
                                                                  int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    const float gm = gamma[c];
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float r = dW2raw[(size_t)c * K + k];
        s += W2[(size_t)c * K + k] * r;
        dW2[(size_t)c * K + k] += gm * r;
    }
    s = wave_sum(s);
    if (lane == 0) {
        dgamma[c] += s + b2[c] * db2raw[c];
        db2[c] += gm * db2raw[c];
    }
}
MMG_API int mmg_layerscale_finalize(const float* W2, const float* b2, const float* gamma, const float* dW2raw,
                                    const float* db2raw, float* dW2, float* db2, float* dgamma, int C, int K,
                                    hipStream_t stream) {
    MMG_CHECK_ARG(W2 && b2 && gamma && dW2raw && db2raw && dW2 && db2 && dgamma && C > 0 && K > 0,
                  "mmg_layerscale_finalize: bad argument");
    hipLaunchKernelGGL(layerscale_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, stream, W2, b2, gamma, dW2raw, db2raw, dW2,
                       db2, dgamma, C, K);
    MMG_LAUNCH_CHECK("mmg_layerscale_finalize");
    return 0;
}

// dst[i] += src[perm(i)] for the small layout changes between GEMM-shaped gradients and torch's conv layouts:
//   mode 0: src [R, KH*KW*CI] (kh,kw,ci fastest) -> dst [R, CI, KH, KW]     (patchify convolutions)
//   mode 1: src [49, C]                          -> dst [C, 49]             (depthwise taps)
__global__ __launch_bounds__(256) void grad_relayout_kernel(const float* __restrict__ src, float* __restrict__ dst, int mode,
                                                            int R, int CI, int KH, int KW, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (mode == 0) {
            const int kw = (int)(i % KW), kh = (int)((i / KW) % KH), ci = (int)((i / ((size_t)KW * KH)) % CI);
            const size_t r = i / ((size_t)KW * KH * CI);
            dst[i] += src[r * ((size_t)KH * KW * CI) + ((size_t)kh * KW + kw) * CI + ci];
        } else {
            const int k = (int)(i % 49);
            const size_t c = i / 49;
            dst[i] += src[(size_t)k * R + c];
        }
    }
}
MMG_API int mmg_grad_relayout(const float* src, float* dst, int mode, int R, int CI, int KH, int KW, int ld_src,
                              hipStream_t stream) {
    MMG_CHECK_ARG(src && dst && (mode == 0 || mode == 1) && R > 0, "mmg_grad_relayout: bad argument");
    MMG_CHECK_ARG(mode == 1 || ld_src == KH * KW * CI, "mmg_grad_relayout: ld_src=%d must equal KH*KW*CI", ld_src);
    const size_t n = mode == 0 ? (size_t)R * CI * KH * KW : (size_t)R * 49;
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(grad_relayout_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, mode, R, CI, KH, KW, n);
    MMG_LAUNCH_CHECK("mmg_grad_relayout");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Dropout for the projection heads (MultiLinearHead / MLPProjectionHead, mmgclip/networks/projection.py:50,59,91,98).
// Counter-based generator (one 64-bit splitmix round per element, keyed by seed): reproducible for a seed, not
// bit-compatible with torch's Philox stream (parity runs use p = 0, SURVEY.md §7 "Hard parts").
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ keep, size_t n, float p,
                                                          unsigned long long seed) {
    const float inv = 1.0f / (1.0f - p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float u = (float)(splitmix64(seed ^ (i * 0xD1342543DE82EF95ull)) >> 40) * (1.0f / 16777216.0f);
        const unsigned char k = u >= p;
        keep[i] = k;
        y[i] = k ? x[i] * inv : 0.f;
    }
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ keep,
                                                          float* __restrict__ dx, size_t n, float p) {
    const float inv = 1.0f / (1.0f - p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dx[i] = keep[i] ? dy[i] * inv : 0.f;
}
MMG_API int mmg_dropout_fwd(const float* x, float* y, void* keep, long long n, float p, long long seed, hipStream_t stream) {
    MMG_CHECK_ARG(x && y && keep && n > 0 && p >= 0.f && p < 1.f, "mmg_dropout_fwd: bad argument (p=%f)", p);
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, y, (unsigned char*)keep, (size_t)n, p,
                       (unsigned long long)seed);
    MMG_LAUNCH_CHECK("mmg_dropout_fwd");
    return 0;
}
MMG_API int mmg_dropout_bwd(const float* dy, const void* keep, float* dx, long long n, float p, hipStream_t stream) {
    MMG_CHECK_ARG(dy && dx && keep && n > 0 && p >= 0.f && p < 1.f, "mmg_dropout_bwd: bad argument");
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dy, (const unsigned char*)keep, dx, (size_t)n, p);
    MMG_LAUNCH_CHECK("mmg_dropout_bwd");
    return 0;
}

// out = dy * act'(pre) elementwise (bf16), kind 0 = GELU(erf), 1 = ReLU, 2 = GELU(erf) by the exp-free polynomial of the on-chip
// weight-gradient backward (common.h: gelu_bf16_grad_poly; here for its all-bf16-inputs test); n % 8 == 0
__global__ __launch_bounds__(256) void act_grad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ pre,
                                                       bf16_t* __restrict__ out, size_t nvec, int kind) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        float g[8], h[8];
        unpack8(reinterpret_cast<const uint4*>(dy)[i], g);
        unpack8(reinterpret_cast<const uint4*>(pre)[i], h);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = kind == 0 ? g[e] * gelu_bf16_grad(h[e]) : (kind == 2 ? g[e] * gelu_bf16_grad_poly(h[e]) : (h[e] > 0.f ? g[e] : 0.f));
        reinterpret_cast<uint4*>(out)[i] = pack8(g);
    }
}
MMG_API int mmg_act_grad_bf16(const void* dy, const void* pre, void* out, long long n, int kind, hipStream_t stream) {
    MMG_CHECK_ARG(dy && pre && out && n > 0 && n % 8 == 0 && (kind >= 0 && kind <= 2), "mmg_act_grad_bf16: bad argument");
    const size_t nvec = (size_t)n / 8;
    int blocks = (int)((nvec + 255) / 256 > 8192 ? 8192 : (nvec + 255) / 256);
    hipLaunchKernelGGL(act_grad_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)pre, (bf16_t*)out, nvec, kind);
    MMG_LAUNCH_CHECK("mmg_act_grad_bf16");
    return 0;
}
