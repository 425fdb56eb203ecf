// Depthwise 7x7 convolution (padding 3, stride 1) in NHWC bf16 for the ConvNeXt blocks (gfx950).
//
// Replaces torchvision CNBlock's Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim, bias=True) inside the
// `features` of the reference's ConvNeXt archive (mmgclip/networks/encoder.py:53,
// mmgclip/networks/image_features.py:100; module tree in notebooks/clf_convnext_tiny_experimental.ipynb cell 3).
//
// Not a GEMM: 49 MAC per output and no channel reduction, so it runs on the fp32 VALU with the tile in LDS.
// A workgroup owns an 8x32 pixel tile of a 32-channel slab: the (8+6)x(32+6) halo tile is staged once
// (16-byte coalesced loads, zero padded), a lane owns a channel pair and a strip of 8 output pixels along W and
// slides the 7-tap window over a row of 14 inputs held in registers (56 packed FMAs per 14 LDS reads).
// The same kernel with the taps flipped is the data gradient; the weight gradient keeps the 49 taps of its channel
// pair in registers across tiles and images and reduces through LDS into one atomic per tap per workgroup.
#include "common.h"
#include <stdlib.h>

#define DW_TW 32
#define DW_TH 8
#define DW_CB 32
#define DW_COLS (DW_TW + 6)
#define DW_ROWS (DW_TH + 6)
#define DW_ROWD (DW_COLS * (DW_CB / 2) + 16)   // dwords per LDS row; +16 keeps rows r, r+1 on disjoint bank halves
// DW_TH = 8 keeps the tile at 34 KiB so four workgroups share a CU (latency hiding for the LDS-read / FMA phases)

// Addressing.  The kernels are bound by VALU ISSUE (profiles/r02_dwconv_pmc.txt: 53 % of their VALU instructions were not FMAs), and
// 64-bit pixel addresses cost quarter-rate v_mul_lo_u32 / v_mad_u64_u32 chains per load and store.  So: the image base
// (n, channel slab) is uniform and stays in scalar registers; a lane adds a 32-BIT byte offset built from 24-bit multiplies
// (v_mad_u32_u24, full rate) - the host checks H * W <= 2^24 pixels and H * W * C * 2 < 2^32 bytes per image.
// a * b + c on the 24-bit multiplier (one full-rate instruction; hipcc turns __umul24(a, b) + c into the quarter-rate 64-bit
// v_mad_u64_u32 when it cannot prove the operand ranges)
__device__ __forceinline__ unsigned dw_mad24(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned dw_byte_off(int pix, int C, int chan) {          // ((pix * C) + chan) * 2
    return dw_mad24((unsigned)pix, (unsigned)C, (unsigned)chan) << 1;
}
template <typename T>
__device__ __forceinline__ const T* dw_at(const bf16_t* img, unsigned byte_off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(img) + byte_off);
}

// stage the halo tile (tile origin h0,w0) of the image / channel slab at `img` into LDS, zero outside the image.
// All of a lane's 16-byte loads are issued before the first LDS store: one memory latency per tile instead of one per
// chunk (the chunk-by-chunk loop made staging 3x longer than the 49-tap arithmetic).  Chunk it * 256 + tid is pixel
// (it * 64 + tid / 4) of the 14 x 38 halo tile: walked as (row, col) += (1, 26) with a carry - no division.
template <int ROWD = DW_ROWD, int ROWS = DW_ROWS>
__device__ __forceinline__ void dw_stage(const bf16_t* __restrict__ img, unsigned* tile, int H, int W, int C, int h0, int w0) {
    constexpr int CHUNKS = ROWS * DW_COLS * (DW_CB / 8);
    constexpr int ITERS = (CHUNKS + 255) / 256;
    static_assert(DW_CB == 32 && DW_COLS < 64 && 64 - DW_COLS < DW_COLS, "the (row, col) walk assumes 4 chunks per pixel, 38 columns");
    uint4 v[ITERS];
    int dst[ITERS];
    const int ch = threadIdx.x & 3;
    int col = threadIdx.x >> 2, row = 0;
    if (col >= DW_COLS) { col -= DW_COLS; row = 1; }
    const int pix00 = __mul24(h0 - 3, W) + (w0 - 3);                  // uniform; may be negative (then out of bounds below)
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int gh = h0 - 3 + row, gw = w0 - 3 + col;
        v[it] = make_uint4(0, 0, 0, 0);
        dst[it] = row < ROWS ? row * ROWD + col * (DW_CB / 2) + ch * 4 : -1;
        if (row < ROWS && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
            v[it] = *dw_at<uint4>(img, dw_byte_off((int)dw_mad24((unsigned)row, (unsigned)W, (unsigned)(pix00 + col)), C, ch * 8));
        col += 64 - DW_COLS;
        row += 1;
        if (col >= DW_COLS) { col -= DW_COLS; row += 1; }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if (dst[it] >= 0) *reinterpret_cast<uint4*>(tile + dst[it]) = v[it];
}

template <bool FLIP>
__global__ __launch_bounds__(256, 2) void dwconv7_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const bf16_t* __restrict__ add,
                                                         bf16_t* __restrict__ y, int H, int W, int C, int tiles_w, int nt) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
    unsigned* tile = smem_u;                                             // [DW_ROWS][DW_ROWD] dwords (bf16 pairs)
    float* ws = reinterpret_cast<float*>(smem_u + DW_ROWS * DW_ROWD);    // [49][32]
    const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
    const int c0 = blockIdx.y * DW_CB, n = blockIdx.z;
    const int h0 = th * DW_TH, w0 = tw * DW_TW;
    const int cp = threadIdx.x & 15, r4 = (threadIdx.x >> 4) & 3, strip = threadIdx.x >> 6;

    for (int i = threadIdx.x; i < 49 * DW_CB; i += 256) {
        const int k = i / DW_CB, c = i % DW_CB;
        ws[i] = w[(size_t)(FLIP ? 48 - k : k) * C + c0 + c];
    }
    dw_stage(x + (size_t)n * H * W * C + c0, tile, H, W, C, h0, w0);
    __syncthreads();

    const float b0 = bias ? bias[c0 + 2 * cp] : 0.f, b1 = bias ? bias[c0 + 2 * cp + 1] : 0.f;
#pragma unroll 1
    for (int pass = 0; pass < DW_TH / 4; ++pass) {
        const int oh = pass * 4 + r4;
        const int gh = h0 + oh;
        // the residual-gradient operand of this pass is fetched first: its latency hides under the 49-tap loop
        unsigned addv[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int gw = w0 + strip * 8 + p;
            addv[p] = (add && gh < H && gw < W) ? *reinterpret_cast<const unsigned*>(add + (((size_t)n * H + gh) * W + gw) * C + c0 + 2 * cp) : 0u;
        }
        float a0[8], a1[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) { a0[p] = b0; a1[p] = b1; }
        // software pipeline over the 7 kernel rows, two named register sets (A/B) in a ROLLED loop of row pairs: row
        // kh+1 (inputs + taps) is read from LDS while row kh is consumed.  (Fully unrolled, hipcc hoists every row's
        // reads to the top and spills.)
        unsigned inA[14], inB[14];
        float2 wA[7], wB[7];
#define DW_LOAD_ROW(IN, WT, KH)                                                                          \
        {                                                                                                \
            const unsigned* rowp_ = tile + (oh + (KH)) * DW_ROWD + (strip * 8) * (DW_CB / 2) + cp;       \
            _Pragma("unroll") for (int q = 0; q < 14; ++q) IN[q] = rowp_[q * (DW_CB / 2)];               \
            _Pragma("unroll") for (int kw = 0; kw < 7; ++kw)                                             \
                WT[kw] = *reinterpret_cast<const float2*>(ws + ((KH) * 7 + kw) * DW_CB + 2 * cp);        \
        }
#define DW_FMA_ROW(IN, WT)                                                                               \
        _Pragma("unroll") for (int kw = 0; kw < 7; ++kw) {                                               \
            _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                              \
                a0[p] = fmaf(WT[kw].x, bf2f_lo(IN[p + kw]), a0[p]);                                      \
                a1[p] = fmaf(WT[kw].y, bf2f_hi(IN[p + kw]), a1[p]);                                      \
            }                                                                                            \
        }
        DW_LOAD_ROW(inA, wA, 0)
#pragma unroll 1
        for (int kh = 0; kh < 6; kh += 2) {
            DW_LOAD_ROW(inB, wB, kh + 1)
            DW_FMA_ROW(inA, wA)
            DW_LOAD_ROW(inA, wA, kh + 2)
            DW_FMA_ROW(inB, wB)
        }
        DW_FMA_ROW(inA, wA)
#undef DW_LOAD_ROW
#undef DW_FMA_ROW
        if (gh < H) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int gw = w0 + strip * 8 + p;
                if (gw < W) {
                    const size_t off = (((size_t)n * H + gh) * W + gw) * C + c0 + 2 * cp;
                    const unsigned o = pack2bf(a0[p] + bf2f_lo(addv[p]), a1[p] + bf2f_hi(addv[p]));
                    if (nt) __builtin_nontemporal_store(o, reinterpret_cast<unsigned*>(y + off));
                    else *reinterpret_cast<unsigned*>(y + off) = o;
                }
            }
        }
    }
}

// Two output rows per lane.  A lane of the kernel above unpacks every input row it reads (a shift or a mask per bf16: 196 of
// its 980 VALU operations per pass) for ONE output row; here rows 2*r4 and 2*r4+1 share their 8 input rows, so a tile costs
// 224 unpacks + 1568 FMAs per lane instead of 392 + 1568, in one pass.  Every output accumulates its 49 taps in the same
// (kh, kw) order as above: bit-identical results.  The row pitch is 616 dwords so that the four row groups of a wave, now
// two rows apart, still start 16 banks apart.
#define DW_ROWD2 (DW_COLS * (DW_CB / 2) + 8)
#define DW_FWD_TH_DEFAULT 8
// TH = tile height: 8 (four workgroups per CU) or 16 (two passes over one staged 22-row tile, two workgroups per CU: halo 1.21x
// instead of 1.4x, staging and tap loads amortised over twice the pixels).
template <bool FLIP, int TH>
__global__ __launch_bounds__(256, 2) void dwconv7_rows2_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const bf16_t* __restrict__ add,
                                                               bf16_t* __restrict__ y, int H, int W, int C, int tiles_w, int nt) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
    constexpr int ROWS = TH + 6;
    unsigned* tile = smem_u;                                              // [ROWS][DW_ROWD2] dwords (bf16 pairs)
    float* ws = reinterpret_cast<float*>(smem_u + ROWS * DW_ROWD2);       // [49][32]
    const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
    const int c0 = blockIdx.y * DW_CB, n = blockIdx.z;
    const int h0 = th * TH, w0 = tw * DW_TW;
    const int cp = threadIdx.x & 15, r4 = (threadIdx.x >> 4) & 3, strip = threadIdx.x >> 6;

    for (int i = threadIdx.x; i < 49 * DW_CB; i += 256) {
        const int k = i / DW_CB, c = i % DW_CB;
        ws[i] = w[(size_t)(FLIP ? 48 - k : k) * C + c0 + c];
    }
    const size_t slab = (size_t)n * H * W * C + c0;                 // uniform: this image, this channel slab
    dw_stage<DW_ROWD2, ROWS>(x + slab, tile, H, W, C, h0, w0);
    __syncthreads();

    const float b0 = bias ? bias[c0 + 2 * cp] : 0.f, b1 = bias ? bias[c0 + 2 * cp + 1] : 0.f;
    char* yslab = reinterpret_cast<char*>(y + slab);
#pragma unroll 1
    for (int pass = 0; pass < TH / 8; ++pass) {
    const int oh = 8 * pass + 2 * r4;
    // byte offsets of this lane's two output rows (pixel 0 of its strip); pixel p is p * C * 2 bytes further.  A tile that lies
    // inside the image (uniform test) needs no per-pixel bounds checks.
    const bool interior = h0 + TH <= H && w0 + DW_TW <= W;
    const unsigned pstep = (unsigned)C * 2u;
    unsigned boff[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) boff[rr] = dw_byte_off((int)dw_mad24((unsigned)(h0 + oh + rr), (unsigned)W, (unsigned)(w0 + strip * 8)), C, 2 * cp);
    unsigned addv[2][8];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const bool in = interior || (h0 + oh + rr < H && w0 + strip * 8 + p < W);
            addv[rr][p] = (add && in) ? *dw_at<unsigned>(add + slab, boff[rr] + p * pstep) : 0u;
        }
    float a0[2][8], a1[2][8];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int p = 0; p < 8; ++p) { a0[rr][p] = b0; a1[rr][p] = b1; }
    const unsigned* base = tile + oh * DW_ROWD2 + (strip * 8) * (DW_CB / 2) + cp;
    const float* wbase = ws + 2 * cp;
    // input row r (0..7 below the pair's first output row) feeds output row 0 with kernel row r and output row 1 with kernel
    // row r-1.  Rolled loop (hipcc hoists every row's LDS reads to the top and spills when it is fully unrolled).
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
        unsigned in[14];
        float f0[14], f1[14];
#pragma unroll
        for (int q = 0; q < 14; ++q) in[q] = base[r * DW_ROWD2 + q * (DW_CB / 2)];
#pragma unroll
        for (int q = 0; q < 14; ++q) { f0[q] = bf2f_lo(in[q]); f1[q] = bf2f_hi(in[q]); }
        if (r < 7) {
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float2 wt = *reinterpret_cast<const float2*>(wbase + (r * 7 + kw) * DW_CB);
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    a0[0][p] = fmaf(wt.x, f0[p + kw], a0[0][p]);
                    a1[0][p] = fmaf(wt.y, f1[p + kw], a1[0][p]);
                }
            }
        }
        if (r > 0) {
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float2 wt = *reinterpret_cast<const float2*>(wbase + ((r - 1) * 7 + kw) * DW_CB);
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    a0[1][p] = fmaf(wt.x, f0[p + kw], a0[1][p]);
                    a1[1][p] = fmaf(wt.y, f1[p + kw], a1[1][p]);
                }
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (interior || (h0 + oh + rr < H && w0 + strip * 8 + p < W)) {
                const unsigned o = pack2bf(a0[rr][p] + bf2f_lo(addv[rr][p]), a1[rr][p] + bf2f_hi(addv[rr][p]));
                unsigned* dstp = reinterpret_cast<unsigned*>(yslab + (boff[rr] + p * pstep));
                if (nt) __builtin_nontemporal_store(o, dstp);
                else *dstp = o;
            }
        }
    }
    }
}

// dw[k][c] += sum_{n,h,w} x[n, h+kh-3, w+kw-3, c] * dy[n,h,w,c] ;  dbias[c] += sum dy
// A workgroup owns one 32-channel slab and walks (image, tile) work items with a grid stride, keeping the 49 taps x 2
// channels of every lane in registers across items, so the LDS / global atomic reduction is paid once per workgroup
// (it dominated the first version on the small late-stage feature maps).
__global__ __launch_bounds__(256, 1) void dwconv7_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                               float* __restrict__ dw, float* __restrict__ dbias, int N,
                                                               int H, int W, int C, int tiles_w, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
    unsigned* tile = smem_u;
    float* red = reinterpret_cast<float*>(smem_u + DW_ROWS * DW_ROWD);    // [4 waves][50][32]
    const int c0 = blockIdx.y * DW_CB;
    const int cp = threadIdx.x & 15, r4 = (threadIdx.x >> 4) & 3, strip = threadIdx.x >> 6;

    float d0[49], d1[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) { d0[k] = 0.f; d1[k] = 0.f; }
    float sb0 = 0.f, sb1 = 0.f;

    for (int item = blockIdx.x; item < N * tiles; item += gridDim.x) {
        const int n = item / tiles, tl = item - n * tiles;
        const int tw = tl % tiles_w, th = tl / tiles_w;
        const int h0 = th * DW_TH, w0 = tw * DW_TW;
        __syncthreads();
        dw_stage(x + (size_t)n * H * W * C + c0, tile, H, W, C, h0, w0);
        __syncthreads();
#pragma unroll 1
        for (int pass = 0; pass < DW_TH / 4; ++pass) {
            const int oh = pass * 4 + r4;
            const int gh = h0 + oh;
            float g0[8], g1[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int gw = w0 + strip * 8 + p;
                unsigned v = 0;
                if (gh < H && gw < W) v = *reinterpret_cast<const unsigned*>(dy + (((size_t)n * H + gh) * W + gw) * C + c0 + 2 * cp);
                g0[p] = bf2f_lo(v);
                g1[p] = bf2f_hi(v);
                sb0 += g0[p];
                sb1 += g1[p];
            }
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                const unsigned* rowp = tile + (oh + kh) * DW_ROWD + (strip * 8) * (DW_CB / 2) + cp;
                float i0[14], i1[14];
#pragma unroll
                for (int q = 0; q < 14; ++q) {
                    const unsigned v = rowp[q * (DW_CB / 2)];
                    i0[q] = bf2f_lo(v);
                    i1[q] = bf2f_hi(v);
                }
#pragma unroll
                for (int kw = 0; kw < 7; ++kw) {
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        d0[kh * 7 + kw] = fmaf(i0[p + kw], g0[p], d0[kh * 7 + kw]);
                        d1[kh * 7 + kw] = fmaf(i1[p + kw], g1[p], d1[kh * 7 + kw]);
                    }
                }
            }
        }
    }
    // reduce over the four row groups of a wave with shuffles, then over the four waves through LDS (plain stores, no
    // same-address LDS atomics: 16 lanes hammering one word serialised the first version), then one global atomic per tap
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 49; ++k) {
        d0[k] += __shfl_xor(d0[k], 16, 64); d0[k] += __shfl_xor(d0[k], 32, 64);
        d1[k] += __shfl_xor(d1[k], 16, 64); d1[k] += __shfl_xor(d1[k], 32, 64);
    }
    sb0 += __shfl_xor(sb0, 16, 64); sb0 += __shfl_xor(sb0, 32, 64);
    sb1 += __shfl_xor(sb1, 16, 64); sb1 += __shfl_xor(sb1, 32, 64);
    if (r4 == 0) {
        float* rw = red + strip * 50 * DW_CB;
#pragma unroll
        for (int k = 0; k < 49; ++k) *reinterpret_cast<float2*>(rw + k * DW_CB + 2 * cp) = make_float2(d0[k], d1[k]);
        *reinterpret_cast<float2*>(rw + 49 * DW_CB + 2 * cp) = make_float2(sb0, sb1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 50 * DW_CB; i += 256) {
        const float v = red[i] + red[50 * DW_CB + i] + red[2 * 50 * DW_CB + i] + red[3 * 50 * DW_CB + i];
        if (i < 49 * DW_CB) atomicAdd(dw + (size_t)(i / DW_CB) * C + c0 + (i % DW_CB), v);
        else if (dbias) atomicAdd(dbias + c0 + (i - 49 * DW_CB), v);
    }
}

// Weight gradient with two dy rows per lane (same idea as dwconv7_rows2_kernel: the 8 input rows under a row pair are read
// and unpacked once for both).  Input row r meets dy row 0 at kernel row r and dy row 1 at kernel row r-1.
// TH = tile height: 8, or 16 in two passes over the staged tile - the halo then costs (16+6)(32+6) / (16*32) = 1.63 x the tile
// instead of 2.08 x (PMC, round 1: 5.39 GB fetched per launch against 2.55 GB algorithmic = exactly that ratio).
template <int TH>
__global__ __launch_bounds__(256, 1) void dwconv7_wgrad_rows2_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                     float* __restrict__ dw, float* __restrict__ dbias, int N,
                                                                     int H, int W, int C, int tiles_w, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
    unsigned* tile = smem_u;
    constexpr int ROWS = TH + 6;
    float* red = reinterpret_cast<float*>(smem_u + ROWS * DW_ROWD2);    // [4 waves][50][32]
    const int c0 = blockIdx.y * DW_CB;
    const int cp = threadIdx.x & 15, r4 = (threadIdx.x >> 4) & 3, strip = threadIdx.x >> 6;

    float d0[49], d1[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) { d0[k] = 0.f; d1[k] = 0.f; }
    float sb0 = 0.f, sb1 = 0.f;

    for (int item = blockIdx.x; item < N * tiles; item += gridDim.x) {
        const int n = item / tiles, tl = item - n * tiles;
        const int tw = tl % tiles_w, th = tl / tiles_w;
        const int h0 = th * TH, w0 = tw * DW_TW;
        __syncthreads();
        const size_t slab = (size_t)n * H * W * C + c0;             // uniform: this image, this channel slab
        dw_stage<DW_ROWD2, ROWS>(x + slab, tile, H, W, C, h0, w0);
        const bool interior = h0 + TH <= H && w0 + DW_TW <= W;
        const unsigned pstep = (unsigned)C * 2u;
#pragma unroll 1
        for (int pass = 0; pass < TH / 8; ++pass) {
        const int oh = 8 * pass + 2 * r4;
        float g0[2][8], g1[2][8];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const unsigned boff = dw_byte_off((int)dw_mad24((unsigned)(h0 + oh + rr), (unsigned)W, (unsigned)(w0 + strip * 8)), C, 2 * cp);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                unsigned v = 0;
                if (interior || (h0 + oh + rr < H && w0 + strip * 8 + p < W)) v = *dw_at<unsigned>(dy + slab, boff + p * pstep);
                g0[rr][p] = bf2f_lo(v);
                g1[rr][p] = bf2f_hi(v);
                sb0 += g0[rr][p];
                sb1 += g1[rr][p];
            }
        }
        if (pass == 0) __syncthreads();                              // the staged tile is complete (uniform branch)
        const unsigned* base = tile + oh * DW_ROWD2 + (strip * 8) * (DW_CB / 2) + cp;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float i0[14], i1[14];
#pragma unroll
            for (int q = 0; q < 14; ++q) {
                const unsigned v = base[r * DW_ROWD2 + q * (DW_CB / 2)];
                i0[q] = bf2f_lo(v);
                i1[q] = bf2f_hi(v);
            }
            if (r < 7) {
#pragma unroll
                for (int kw = 0; kw < 7; ++kw)
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        d0[r * 7 + kw] = fmaf(i0[p + kw], g0[0][p], d0[r * 7 + kw]);
                        d1[r * 7 + kw] = fmaf(i1[p + kw], g1[0][p], d1[r * 7 + kw]);
                    }
            }
            if (r > 0) {
#pragma unroll
                for (int kw = 0; kw < 7; ++kw)
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        d0[(r - 1) * 7 + kw] = fmaf(i0[p + kw], g0[1][p], d0[(r - 1) * 7 + kw]);
                        d1[(r - 1) * 7 + kw] = fmaf(i1[p + kw], g1[1][p], d1[(r - 1) * 7 + kw]);
                    }
            }
        }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 49; ++k) {
        d0[k] += __shfl_xor(d0[k], 16, 64); d0[k] += __shfl_xor(d0[k], 32, 64);
        d1[k] += __shfl_xor(d1[k], 16, 64); d1[k] += __shfl_xor(d1[k], 32, 64);
    }
    sb0 += __shfl_xor(sb0, 16, 64); sb0 += __shfl_xor(sb0, 32, 64);
    sb1 += __shfl_xor(sb1, 16, 64); sb1 += __shfl_xor(sb1, 32, 64);
    if (r4 == 0) {
        float* rw = red + strip * 50 * DW_CB;
#pragma unroll
        for (int k = 0; k < 49; ++k) *reinterpret_cast<float2*>(rw + k * DW_CB + 2 * cp) = make_float2(d0[k], d1[k]);
        *reinterpret_cast<float2*>(rw + 49 * DW_CB + 2 * cp) = make_float2(sb0, sb1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 50 * DW_CB; i += 256) {
        const float v = red[i] + red[50 * DW_CB + i] + red[2 * 50 * DW_CB + i] + red[3 * 50 * DW_CB + i];
        if (i < 49 * DW_CB) atomicAdd(dw + (size_t)(i / DW_CB) * C + c0 + (i % DW_CB), v);
        else if (dbias) atomicAdd(dbias + c0 + (i - 49 * DW_CB), v);
    }
}

static int dw_check(const char* who, int n, int H, int W, int C) {
    MMG_CHECK_ARG(n > 0 && H > 0 && W > 0 && C > 0 && C % DW_CB == 0 && n <= 65535 && C / DW_CB <= 65535,
                  "%s: n=%d H=%d W=%d C=%d (C must be a multiple of 32)", who, n, H, W, C);
    // 32-bit byte offsets inside one image, 24-bit pixel indices (dw_byte_off)
    MMG_CHECK_ARG((long long)(H + 16) * (W + 64) <= (1LL << 24) && (long long)H * W * C * 2 < (1LL << 32),
                  "%s: one image of %d x %d x %d exceeds the kernels' 32-bit addressing (H * W <= 2^24, H * W * C * 2 < 2^32)", who, H, W, C);
    return 0;
}

// y = dwconv7(x; w, bias) (+ add).  w is tap-major fp32 [49][C] (w[kh*7+kw][c] = weight[c,0,kh,kw]).
// flip != 0 uses the taps reversed: with x := dy this is the gradient w.r.t. the input.
MMG_API int mmg_dwconv7_nhwc(const void* x, const float* w, const float* bias, const void* add, void* y, int n, int H,
                             int W, int C, int flip, hipStream_t stream) {
    MMG_CHECK_ARG(x && w && y, "mmg_dwconv7_nhwc: null pointer");
    if (dw_check("mmg_dwconv7_nhwc", n, H, W, C)) return 1;
    const int tiles_w = cdiv(W, DW_TW), tiles_h = cdiv(H, DW_TH);
    const size_t shm = (size_t)DW_ROWS * DW_ROWD * 4 + 49 * DW_CB * 4;
    const dim3 grid(tiles_w * tiles_h, C / DW_CB, n);
    const int nt = (size_t)n * H * W * C * 2 >= ((size_t)256 << 20);
    const int rows2 = getenv("MMG_DWCONV_ROWS2") ? atoi(getenv("MMG_DWCONV_ROWS2")) : 1;      // read per call (tests / A-B runs)
    if (rows2) {
        // 16-row tiles from MMG_DWCONV_TH=16 (A/B knob; default 8)
        const int th16 = getenv("MMG_DWCONV_TH") ? atoi(getenv("MMG_DWCONV_TH")) == 16 : DW_FWD_TH_DEFAULT == 16;
        const int TH = th16 ? 16 : 8;
        const size_t shm2 = (size_t)(TH + 6) * DW_ROWD2 * 4 + 49 * DW_CB * 4;
        const dim3 grid2(tiles_w * cdiv(H, TH), C / DW_CB, n);
#define DW_LAUNCH2(FL, T)                                                                                          \
        do {                                                                                                       \
            mmg_allow_lds(dwconv7_rows2_kernel<FL, T>, shm2);                                                      \
            hipLaunchKernelGGL((dwconv7_rows2_kernel<FL, T>), grid2, dim3(256), shm2, stream, (const bf16_t*)x, w, bias, \
                               (const bf16_t*)add, (bf16_t*)y, H, W, C, tiles_w, nt);                              \
        } while (0)
        MMG_NOTE_KERNEL("dwconv7_rows2_kernel<%s, %d>", flip ? "true" : "false", TH);
        if (flip) { if (th16) DW_LAUNCH2(true, 16); else DW_LAUNCH2(true, 8); }
        else      { if (th16) DW_LAUNCH2(false, 16); else DW_LAUNCH2(false, 8); }
#undef DW_LAUNCH2
        MMG_LAUNCH_CHECK("mmg_dwconv7_nhwc");
        return 0;
    }
    MMG_NOTE_KERNEL("dwconv7_kernel<%s>", flip ? "true" : "false");
    if (flip) {
        mmg_allow_lds(dwconv7_kernel<true>, shm);
        hipLaunchKernelGGL(dwconv7_kernel<true>, grid, dim3(256), shm, stream, (const bf16_t*)x, w, bias, (const bf16_t*)add,
                           (bf16_t*)y, H, W, C, tiles_w, nt);
    } else {
        mmg_allow_lds(dwconv7_kernel<false>, shm);
        hipLaunchKernelGGL(dwconv7_kernel<false>, grid, dim3(256), shm, stream, (const bf16_t*)x, w, bias, (const bf16_t*)add,
                           (bf16_t*)y, H, W, C, tiles_w, nt);
    }
    MMG_LAUNCH_CHECK("mmg_dwconv7_nhwc");
    return 0;
}

// dw[49][C] += ..., dbias[C] += ...   (fp32, atomics; caller zeroes once per step)
MMG_API int mmg_dwconv7_wgrad(const void* x, const void* dy, float* dw, float* dbias, int n, int H, int W, int C,
                              hipStream_t stream) {
    if (dw_check("mmg_dwconv7_wgrad", n, H, W, C)) return 1;
    MMG_CHECK_ARG(x && dy && dw, "mmg_dwconv7_wgrad: null pointer");
    const int rows2 = getenv("MMG_DWCONV_ROWS2") ? atoi(getenv("MMG_DWCONV_ROWS2")) : 1;
    // tile height of the two-rows-per-lane kernel: 16 where the map is tall enough to fill the persistent grid (halo 1.63x), else 8
    const int th_env = getenv("MMG_DWG_TH") ? atoi(getenv("MMG_DWG_TH")) : 0;
    const int TH = rows2 ? (th_env == 8 || th_env == 16 ? th_env : ((long)n * cdiv(H, 16) * cdiv(W, DW_TW) * (C / DW_CB) >= 2048 ? 16 : 8)) : DW_TH;
    const int tiles_w = cdiv(W, DW_TW), tiles_h = cdiv(H, TH);
    const size_t shm = (size_t)DW_ROWS * DW_ROWD * 4 + 4 * 50 * DW_CB * 4;
    // ~1024 workgroups in total (4 per CU), each walking its share of the (image, tile) items of one channel slab
    const int tiles = tiles_w * tiles_h, slabs = C / DW_CB;
    int per_slab = 1024 / slabs;
    // a multiple of 8: workgroup x of every slab then lands on XCD x % 8 (ids are x + per_slab * slab), so the slabs of one
    // pixel - 64-byte pieces of the same 128-byte lines, walked in the same order - share one L2 (measured before: 2.6x the
    // algorithmic bytes left the L2s)
    static const int align8 = getenv("MMG_DWG_ALIGN") ? atoi(getenv("MMG_DWG_ALIGN")) : 1;
    if (align8 && per_slab >= 8) per_slab &= ~7;
    if (per_slab < 1) per_slab = 1;
    if (per_slab > n * tiles) per_slab = n * tiles;
    if (rows2) {
        const size_t shm2 = (size_t)(TH + 6) * DW_ROWD2 * 4 + 4 * 50 * DW_CB * 4;
        MMG_NOTE_KERNEL("dwconv7_wgrad_rows2_kernel<%d>", TH);
        if (TH == 16) {
            mmg_allow_lds(dwconv7_wgrad_rows2_kernel<16>, shm2);
            hipLaunchKernelGGL(dwconv7_wgrad_rows2_kernel<16>, dim3(per_slab, slabs), dim3(256), shm2, stream, (const bf16_t*)x,
                               (const bf16_t*)dy, dw, dbias, n, H, W, C, tiles_w, tiles);
        } else {
            mmg_allow_lds(dwconv7_wgrad_rows2_kernel<8>, shm2);
            hipLaunchKernelGGL(dwconv7_wgrad_rows2_kernel<8>, dim3(per_slab, slabs), dim3(256), shm2, stream, (const bf16_t*)x,
                               (const bf16_t*)dy, dw, dbias, n, H, W, C, tiles_w, tiles);
        }
        MMG_LAUNCH_CHECK("mmg_dwconv7_wgrad");
        return 0;
    }
    MMG_NOTE_KERNEL("dwconv7_wgrad_kernel");
    mmg_allow_lds(dwconv7_wgrad_kernel, shm);
    hipLaunchKernelGGL(dwconv7_wgrad_kernel, dim3(per_slab, slabs), dim3(256), shm, stream, (const bf16_t*)x,
                       (const bf16_t*)dy, dw, dbias, n, H, W, C, tiles_w, tiles);
    MMG_LAUNCH_CHECK("mmg_dwconv7_wgrad");
    return 0;
}
