// Depthwise 7x7 convolution on the matrix cores (gfx950) - forward and data gradient of torchvision CNBlock's
// Conv2d(dim, dim, 7, padding=3, groups=dim) (reference: mmgclip/networks/encoder.py:53 runs the ConvNeXt `features`).
//
// A depthwise conv has no channel reduction, so it is not a GEMM - but for ONE channel and ONE kernel row ky the 1-D
// convolution along x of 16 output columns is a product with a banded Toeplitz matrix:
//     out[y][x0+n] += sum_k in[y+ky-3][x0-3+k] * T_ky[k][n],   T_ky[k][n] = w[ky][k-n] if 0 <= k-n <= 6 else 0,  k < 32
// i.e. one v_mfma_f32_16x16x32_bf16 per (channel, ky) and 16x16 output pixels: A = 16 input rows x 32 input columns,
// B = T_ky.  22 of the 32 k carry data (49*256 useful MACs of 7*8192 = 22 %), which at the MFMA rate is still ~8x the fp32
// VALU rate the direct form is limited to (16 FMA lanes/clk/SIMD).
//
// What it costs is a layout change: the MFMA contracts over x, NHWC has c fastest.  A workgroup owns a 16x16 pixel tile of a
// 32-channel slab; the 22x22 halo tile is loaded as NHWC 16-byte pieces (64 contiguous bytes per pixel) and scattered into
// channel-planar LDS ([c][row][x], rows 80 B apart so that 16 rows x 16 B never share a bank), zero padded.  A compute wave
// owns 4 channels and keeps their 28 Toeplitz fragments (taps rounded to bf16) in registers for its whole life (the workgroup
// is persistent over the spatial tiles of its slab), so the inner loop is one ds_read_b128 + one MFMA.
//
// STATUS (round 1, measured at 16 x 256 x 256 x 96): correct within bf16 tolerance (tests), forward 206-248 us against 214 us
// for the direct fp32-VALU kernel, data gradient 282 us against 275 us - parity, not a win, so the direct kernel stays the
// default (this one: MMG_DWCONV_MFMA=1 or mmg_dwconv7_nhwc_mfma).  Ablation of the 16-channel-slab variant: MFMAs 30 us, input
// loads 90 us, output stores 124 us of 247 us - the matrix cores are idle; what bounds it is the number of memory requests
// (a slab's piece of a pixel is 32-64 B of a 128-byte line) and the 2-byte/4-byte LDS scatter of the layout change.
#include "common.h"
#include <stdlib.h>

#define DM_TH 16
#define DM_TW 16
#define DM_CB 32                            // channels per workgroup: 64 contiguous bytes per pixel (4 lanes x 16 B)
#define DM_ROWS (DM_TH + 6)                 // 22
#define DM_COLS (DM_TW + 6)                 // 22 columns carry data; columns 22..31 are read (times zero) and stay zero
#define DM_ROWB 80                          // bytes per plane row (40 bf16 >= the 32 k of one MFMA); 16 rows x 16 B hit 16 bank groups
#define DM_PLANE (DM_ROWS * DM_ROWB + 16)
#define DM_PLANES (DM_CB * DM_PLANE)         // one set of channel planes: 56 832 B
#define DM_OUTB (DM_TH * DM_TW * DM_CB * 2)  // one NHWC staging tile: 16 KiB
#define DM_LDS (2 * DM_PLANES + 2 * DM_OUTB)
#define DM_THREADS 768

struct DwMfma {
    const bf16_t* x; const float* w; const float* bias; const bf16_t* add; bf16_t* y;
    int n, H, W, C, tiles_w, tiles_h, splits, xcd_grouped;
};

// 12 waves: waves 0-7 compute (4 channels each, Toeplitz fragments resident: 7 MFMAs per channel and tile, results packed
// four channels at a time into an NHWC staging tile), waves 8-11 move data - they scatter the halo tile of item k+1
// (loaded into registers one iteration earlier, pairs of x-adjacent pixels -> one 4-byte LDS store per channel) into the
// other set of planes, issue the loads of item k+2 and write item k-1's staged results to HBM as 16-byte pieces (+ the
// residual-gradient add).  Two sets of planes and two staging tiles: ONE barrier per tile, no memory latency on the MFMA path.
template <bool FLIP>
__global__ __launch_bounds__(DM_THREADS, 1) void dwconv7_mfma_kernel(const DwMfma p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    // The 32-channel slabs of one pixel share 128-byte lines: the workgroups that walk the same spatial items for the different
    // slabs sit on the SAME XCD (hardware: XCD = workgroup id % 8) so the line is fetched into one L2, not one per slab.
    const int slabs = p.C / DM_CB;
    int bx, c0;                                            // which stream of spatial items, which slab
    if (p.xcd_grouped) {
        const int L = blockIdx.x, xcd = L & 7, i = L >> 3;
        bx = (i / slabs) * 8 + xcd;
        c0 = (i % slabs) * DM_CB;
    } else {
        bx = blockIdx.x / slabs;
        c0 = (blockIdx.x % slabs) * DM_CB;
    }
    if (bx >= p.splits) return;
    const bool mover = wave >= 8;
    const int per_img = p.tiles_w * p.tiles_h, items = p.n * per_img;
    const int my_items = (bx < items) ? (items - 1 - bx) / p.splits + 1 : 0;
    auto origin = [&](int k, int& img, int& Y0, int& X0) {
        const int item = bx + k * p.splits;
        img = item / per_img;
        const int t = item - img * per_img, ty = t / p.tiles_w;
        Y0 = ty * DM_TH; X0 = (t - ty * p.tiles_w) * DM_TW;
    };

    for (int i = tid; i < 2 * DM_PLANES / 16; i += DM_THREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    if (!mover) {
        // ---- Toeplitz fragments of this wave's 4 channels: B[k = 8 lg + j][n = li] = tap[ky][k - n] ---------------------------
        bf16x8 T[4][7];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int c = c0 + 4 * wave + ch;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                unsigned pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int d = 8 * lg + 2 * e + h - li;
                        const int tap = FLIP ? (6 - ky) * 7 + (6 - d) : ky * 7 + d;
                        v[h] = (d >= 0 && d <= 6) ? p.w[(size_t)tap * p.C + c] : 0.f;
                    }
                    pk[e] = pack2bf(v[0], v[1]);
                }
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                T[ch][ky] = __builtin_bit_cast(bf16x8, (u32x4_t{pk[0], pk[1], pk[2], pk[3]}));
            }
        }
        float bia[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) bia[ch] = p.bias ? p.bias[c0 + 4 * wave + ch] : 0.f;

        __syncthreads();                                   // item 0 staged
        for (int k = 0; k < my_items; ++k) {
            const char* planes = smem + (k & 1) * DM_PLANES + (4 * wave) * DM_PLANE + li * DM_ROWB + lg * 16;
            char* outt = smem + 2 * DM_PLANES + (k & 1) * DM_OUTB;
            f32x4 acc[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) acc[ch] = f32x4{bia[ch], bia[ch], bia[ch], bia[ch]};
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                bf16x8 in[4];
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) in[ch] = *reinterpret_cast<const bf16x8*>(planes + ch * DM_PLANE + ky * DM_ROWB);
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) acc[ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(in[ch], T[ch][ky], acc[ch], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);        // keep at most one row of fragments in flight (112 registers hold T)
            }
            // accumulators (lane: column li, rows 4 lg + r) -> NHWC staging tile, 4 channels = 8 bytes per pixel
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<uint2*>(outt + (((4 * lg + r) * DM_TW + li) * DM_CB + 4 * wave) * 2) =
                    make_uint2(pack2bf(acc[0][r], acc[1][r]), pack2bf(acc[2][r], acc[3][r]));
            __syncthreads();
        }
    } else {
        const int lt = tid - 512;
        constexpr int PPR = DM_COLS / 2;                                             // pixel pairs per halo row
        constexpr int PAIRS = DM_ROWS * PPR * 4, ITERS = (PAIRS + 255) / 256;         // 968 -> 4 iterations
        uint4 v[ITERS][2];
        auto fetch = [&](int k) {                          // halo tile of item k -> registers (16-byte pieces, 64 B per pixel)
            int img, Y0, X0;
            origin(k, img, Y0, X0);
            const bf16_t* xi = p.x + (size_t)img * p.H * p.W * p.C;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = it * 256 + lt;
                const int g = idx & 3, pr = idx >> 2;
                const int r = pr / PPR, col = 2 * (pr - r * PPR);
                const int gy = Y0 - 3 + r, gx = X0 - 3 + col;
                const bool rok = idx < PAIRS && gy >= 0 && gy < p.H;
                const bf16_t* src = xi + ((size_t)gy * p.W + gx) * p.C + c0 + 8 * g;
                v[it][0] = (rok && gx >= 0 && gx < p.W) ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
                v[it][1] = (rok && gx + 1 >= 0 && gx + 1 < p.W) ? *reinterpret_cast<const uint4*>(src + p.C) : make_uint4(0, 0, 0, 0);
            }
        };
        auto scatter = [&](char* planes) {                 // registers -> channel planes, two pixels per 4-byte store
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = it * 256 + lt;
                if (idx < PAIRS) {
                    const int g = idx & 3, pr = idx >> 2;
                    const int r = pr / PPR, col = 2 * (pr - r * PPR);
                    char* base = planes + (8 * g) * DM_PLANE + r * DM_ROWB + col * 2;
                    const unsigned a[4] = {v[it][0].x, v[it][0].y, v[it][0].z, v[it][0].w};
                    const unsigned b[4] = {v[it][1].x, v[it][1].y, v[it][1].z, v[it][1].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        *reinterpret_cast<unsigned*>(base + (2 * e) * DM_PLANE) = (a[e] & 0xffffu) | (b[e] << 16);
                        *reinterpret_cast<unsigned*>(base + (2 * e + 1) * DM_PLANE) = (a[e] >> 16) | (b[e] & 0xffff0000u);
                    }
                }
            }
        };
        auto drain = [&](int k) {                          // staged results of item k -> HBM (+ residual-gradient add)
            int img, Y0, X0;
            origin(k, img, Y0, X0);
            const char* outt = smem + 2 * DM_PLANES + (k & 1) * DM_OUTB;
            bf16_t* yo = p.y + (size_t)img * p.H * p.W * p.C;
            const bf16_t* ad = p.add ? p.add + (size_t)img * p.H * p.W * p.C : nullptr;
            constexpr int DIT = (DM_TH * DM_TW * 4) / 256;
            uint4 av[DIT];
            if (ad) {
#pragma unroll
                for (int it = 0; it < DIT; ++it) {
                    const int idx = it * 256 + lt, g = idx & 3, pix = idx >> 2;
                    const int gy = Y0 + pix / DM_TW, gx = X0 + pix % DM_TW;
                    av[it] = (gy < p.H && gx < p.W) ? *reinterpret_cast<const uint4*>(ad + ((size_t)gy * p.W + gx) * p.C + c0 + 8 * g)
                                                    : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const int idx = it * 256 + lt, g = idx & 3, pix = idx >> 2;
                const int gy = Y0 + pix / DM_TW, gx = X0 + pix % DM_TW;
                if (gy < p.H && gx < p.W) {
                    uint4 o = *reinterpret_cast<const uint4*>(outt + (pix * DM_CB + 8 * g) * 2);
                    if (ad) {
                        const uint4 a = av[it];
                        o.x = pack2bf(bf2f_lo(o.x) + bf2f_lo(a.x), bf2f_hi(o.x) + bf2f_hi(a.x));
                        o.y = pack2bf(bf2f_lo(o.y) + bf2f_lo(a.y), bf2f_hi(o.y) + bf2f_hi(a.y));
                        o.z = pack2bf(bf2f_lo(o.z) + bf2f_lo(a.z), bf2f_hi(o.z) + bf2f_hi(a.z));
                        o.w = pack2bf(bf2f_lo(o.w) + bf2f_lo(a.w), bf2f_hi(o.w) + bf2f_hi(a.w));
                    }
                    *reinterpret_cast<uint4*>(yo + ((size_t)gy * p.W + gx) * p.C + c0 + 8 * g) = o;
                }
            }
        };
        if (my_items > 0) { fetch(0); scatter(smem); }
        if (my_items > 1) fetch(1);
        __syncthreads();                                   // item 0 staged
        for (int k = 0; k < my_items; ++k) {
            if (k + 1 < my_items) scatter(smem + ((k + 1) & 1) * DM_PLANES);     // loaded during the previous iteration
            if (k + 2 < my_items) fetch(k + 2);
            if (k >= 1) drain(k - 1);
            __syncthreads();
        }
        if (my_items > 0) drain(my_items - 1);
    }
}

static int dwconv7_mfma_launch_impl(const void* x, const float* w, const float* bias, const void* add, void* y, int n, int H, int W, int C,
                        int flip, hipStream_t stream) {
    DwMfma p{(const bf16_t*)x, w, bias, (const bf16_t*)add, (bf16_t*)y, n, H, W, C, cdiv(W, DM_TW), cdiv(H, DM_TH), 0, 0};
    const int slabs = C / DM_CB;
    const long items = (long)n * p.tiles_w * p.tiles_h;
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t pr;
        (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&pr, dev);
        cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    // one 8-wave workgroup per CU and never a second round of workgroups (a few extra workgroups would double the run time)
    static const int grouped = getenv("MMG_DWCONV_XCD") ? atoi(getenv("MMG_DWCONV_XCD")) : 1;
    const int per_xcd = (cus / 8) / slabs;               // item streams whose slabs all fit one XCD
    int splits;
    if (grouped && per_xcd >= 1) { splits = 8 * per_xcd; p.xcd_grouped = 1; }
    else { splits = cus / slabs; p.xcd_grouped = 0; }
    if (splits > items) splits = (int)items;
    if (splits < 1) splits = 1;
    p.splits = splits;
    const int grid = p.xcd_grouped ? 8 * ((splits + 7) / 8) * slabs : splits * slabs;
    if (flip) {
        mmg_allow_lds(dwconv7_mfma_kernel<true>, DM_LDS);
        hipLaunchKernelGGL(dwconv7_mfma_kernel<true>, dim3(grid), dim3(DM_THREADS), DM_LDS, stream, p);
    } else {
        mmg_allow_lds(dwconv7_mfma_kernel<false>, DM_LDS);
        hipLaunchKernelGGL(dwconv7_mfma_kernel<false>, dim3(grid), dim3(DM_THREADS), DM_LDS, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// returns 0 on success; arguments validated by the callers
int dwconv7_mfma_launch(const void* x, const float* w, const float* bias, const void* add, void* y, int n, int H, int W, int C,
                        int flip, hipStream_t stream) {
    return dwconv7_mfma_launch_impl(x, w, bias, add, y, n, H, W, C, flip, stream);
}

MMG_API int mmg_dwconv7_nhwc_mfma(const void* x, const float* w, const float* bias, const void* add, void* y, int n, int H,
                                  int W, int C, int flip, hipStream_t stream) {
    MMG_CHECK_ARG(x && w && y, "mmg_dwconv7_nhwc_mfma: null pointer");
    MMG_CHECK_ARG(n > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0, "mmg_dwconv7_nhwc_mfma: n=%d H=%d W=%d C=%d (C must be a multiple of 32)",
                  n, H, W, C);
    const int rc = dwconv7_mfma_launch_impl(x, w, bias, add, y, n, H, W, C, flip, stream);
    if (rc) mmg_set_error("mmg_dwconv7_nhwc_mfma: launch failed");
    return rc;
}
