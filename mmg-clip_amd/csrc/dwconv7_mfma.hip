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
// STATUS (round 1, measured): bit-compatible with the direct kernel within bf16 tolerance and as fast in the forward
// (206-226 us vs 214 us at 16 x 256 x 256 x 96), slower with the residual add; the MFMA loop itself is 12 % of the run time.
// What bounds it is the texture-address path, not the matrix cores: a 16-channel slab is 32 B per pixel, so every 16-byte
// lane access is its own 128-byte line (ablation: input loads 90 us, 8-byte output stores 124 us, MFMAs 30 us of 247 us).
// A 64-channel slab would make the accesses whole lines but needs 224 registers of Toeplitz fragments per wave.  The
// direct kernel therefore stays the default; this one is selected with MMG_DWCONV_MFMA=1 / mmg_dwconv7_nhwc_mfma.
//
// What it costs is a layout change: the MFMA contracts over x, NHWC has c fastest.  A workgroup (4 waves) owns a 16x32
// pixel tile of a 16-channel slab; the 22x38 halo tile is loaded as NHWC 16-byte pieces and scattered into channel-planar
// LDS ([c][row][x], rows 112 B apart so that 16 rows x 16 B never share a bank), zero padded.  A wave owns 4 channels and
// keeps their 28 Toeplitz fragments (taps rounded to bf16) in registers for its whole life (the workgroup is persistent
// over the spatial tiles of its slab), so the inner loop is one ds_read_b128 + one MFMA.  The accumulators (lane = output
// column, 4 rows) are packed four channels at a time into an NHWC staging tile in LDS and leave as 16-byte stores with
// bias and the optional residual-gradient add fused.
#include "common.h"
#include <stdlib.h>

#define DM_TH 16
#define DM_TW 32
#define DM_CB 16
#define DM_ROWS (DM_TH + 6)                 // 22
#define DM_COLS (DM_TW + 6)                 // 38 columns carry data; columns 38..47 are read (times zero) and stay zero
#define DM_ROWB 112                         // bytes per plane row (56 bf16)
#define DM_PLANE (DM_ROWS * DM_ROWB + 16)   // +16 B: planes 8 apart land on different banks
#define DM_PLANES (DM_CB * DM_PLANE)         // one set of channel planes: 39 680 B
#define DM_LDS (2 * DM_PLANES)

struct DwMfma {
    const bf16_t* x; const float* w; const float* bias; const bf16_t* add; bf16_t* y;
    int n, H, W, C, tiles_w, tiles_h, splits, xcd_grouped;
};

// 8 waves: waves 0-3 compute (4 channels each, Toeplitz fragments resident) and store their accumulators straight to HBM
// (8 bytes = 4 channels per pixel, merged into lines by L2); waves 4-7 move data - they scatter the halo tile of item k+1
// (loaded into registers one iteration earlier) into the other set of planes and issue the loads of item k+2.  Two sets of
// planes: ONE barrier per tile, and no input latency on the MFMA waves' path.
template <bool FLIP>
__global__ __launch_bounds__(512, 1) void dwconv7_mfma_kernel(const DwMfma p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lg = lane >> 4;
    // The 16-channel slabs of one pixel share 128-byte lines (a slab's piece is 32 B): the workgroups that walk the same
    // spatial items for the different slabs must sit on the SAME XCD (hardware: XCD = workgroup id % 8), or every XCD's L2
    // fetches the whole line for its 32 bytes (measured: 4x the algorithmic bytes over the fabric, kernel fabric-bound).
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
    const bool mover = wave >= 4;
    const int per_img = p.tiles_w * p.tiles_h, items = p.n * per_img;
    const int my_items = (bx < items) ? (items - 1 - bx) / p.splits + 1 : 0;

    for (int i = tid; i < 2 * DM_PLANES / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
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
            const char* planes = smem + (k & 1) * DM_PLANES;
            const int item = bx + k * p.splits;
            const int img = item / per_img, t = item - img * per_img, ty = t / p.tiles_w;
            const int Y0 = ty * DM_TH, X0 = (t - ty * p.tiles_w) * DM_TW;
            // lane (li, lg) ends up with output column X0 + 16 sub + li, rows Y0 + 4 lg + r, this wave's 4 channels (8 bytes)
            const size_t pix0 = ((size_t)img * p.H + Y0 + 4 * lg) * p.W + X0 + li;
            uint2 addv[2][4];
            if (p.add) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (Y0 + 4 * lg + r < p.H) && (X0 + 16 * sub + li < p.W);
                        addv[sub][r] = ok ? *reinterpret_cast<const uint2*>(p.add + (pix0 + (size_t)r * p.W + 16 * sub) * p.C + c0 + 4 * wave)
                                          : make_uint2(0, 0);
                    }
            }
            // 4 channels x 2 column blocks x 7 kernel rows: A = plane rows ky .. ky+15, 32 columns from 16*sub
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x4 acc[4];
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) acc[ch] = f32x4{bia[ch], bia[ch], bia[ch], bia[ch]};
#pragma unroll
                for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const bf16x8 in = *reinterpret_cast<const bf16x8*>(planes + (4 * wave + ch) * DM_PLANE + (li + ky) * DM_ROWB +
                                                                          lg * 16 + sub * 32);
                        acc[ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(in, T[ch][ky], acc[ch], 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if ((Y0 + 4 * lg + r < p.H) && (X0 + 16 * sub + li < p.W)) {
                        float o[4] = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                        if (p.add) {
                            o[0] += bf2f_lo(addv[sub][r].x); o[1] += bf2f_hi(addv[sub][r].x);
                            o[2] += bf2f_lo(addv[sub][r].y); o[3] += bf2f_hi(addv[sub][r].y);
                        }
                        *reinterpret_cast<uint2*>(p.y + (pix0 + (size_t)r * p.W + 16 * sub) * p.C + c0 + 4 * wave) =
                            make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // movers: halo tile of item k+1 -> the other set of planes.  A lane owns PAIRS of x-adjacent pixels of one 8-channel
        // group, so each channel leaves as one 4-byte LDS store (two pixels) instead of two 2-byte ones.
        const int lt = tid - 256;
        constexpr int PAIRS = DM_ROWS * (DM_COLS / 2) * 2, ITERS = (PAIRS + 255) / 256;     // 836 -> 4 iterations
        uint4 v[ITERS][2];
        auto fetch = [&](int k) {
            const int item = bx + k * p.splits;
            const int img = item / per_img, t = item - img * per_img, ty = t / p.tiles_w;
            const int Y0 = ty * DM_TH, X0 = (t - ty * p.tiles_w) * DM_TW;
            const bf16_t* xi = p.x + (size_t)img * p.H * p.W * p.C;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = it * 256 + lt;
                const int g = idx & 1, pr = idx >> 1;
                const int r = pr / (DM_COLS / 2), col = 2 * (pr - r * (DM_COLS / 2));
                const int gy = Y0 - 3 + r, gx = X0 - 3 + col;
                const bool rok = idx < PAIRS && gy >= 0 && gy < p.H;
                const bf16_t* src = xi + ((size_t)gy * p.W + gx) * p.C + c0 + 8 * g;
                v[it][0] = (rok && gx >= 0 && gx < p.W) ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
                v[it][1] = (rok && gx + 1 >= 0 && gx + 1 < p.W) ? *reinterpret_cast<const uint4*>(src + p.C) : make_uint4(0, 0, 0, 0);
            }
        };
        auto scatter = [&](char* planes) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = it * 256 + lt;
                if (idx < PAIRS) {
                    const int g = idx & 1, pr = idx >> 1;
                    const int r = pr / (DM_COLS / 2), col = 2 * (pr - r * (DM_COLS / 2));
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
        if (my_items > 0) { fetch(0); scatter(smem); }
        if (my_items > 1) fetch(1);
        __syncthreads();                                   // item 0 staged
        for (int k = 0; k < my_items; ++k) {
            if (k + 1 < my_items) scatter(smem + ((k + 1) & 1) * DM_PLANES);     // loaded during the previous iteration
            if (k + 2 < my_items) fetch(k + 2);
            __syncthreads();
        }
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
        hipLaunchKernelGGL(dwconv7_mfma_kernel<true>, dim3(grid), dim3(512), DM_LDS, stream, p);
    } else {
        mmg_allow_lds(dwconv7_mfma_kernel<false>, DM_LDS);
        hipLaunchKernelGGL(dwconv7_mfma_kernel<false>, dim3(grid), dim3(512), DM_LDS, stream, p);
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
    MMG_CHECK_ARG(n > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0, "mmg_dwconv7_nhwc_mfma: n=%d H=%d W=%d C=%d (C must be a multiple of 16)",
                  n, H, W, C);
    const int rc = dwconv7_mfma_launch_impl(x, w, bias, add, y, n, H, W, C, flip, stream);
    if (rc) mmg_set_error("mmg_dwconv7_nhwc_mfma: launch failed");
    return rc;
}
