// ResNet-50 image tower pieces (NHWC bf16) - im2col / col2im around the MFMA GEMMs, batch norm, 3x3/2 max pool.
//
// Replaces torchvision's resnet50 inside the reference's ResNet50Encoder (mmgclip/networks/encoder.py:57-119: conv1, bn1,
// relu, maxpool, layer1..4, avgpool; every parameter frozen except layer4, :88-89).  Convolutions are GEMMs here:
// 1x1 convolutions are plain row-major GEMMs on NHWC rows, k x k ones go through an explicit column matrix
// (row = output pixel, columns ordered (kh, kw, c) like the relaid-out weight).  This tower is small next to the ConvNeXt
// path (224^2 inputs, or the degenerate 1 x 768 "image" the reference builds from precomputed features, :101-103), so the
// kernels are straightforward streaming kernels: 16 bytes per lane, read once / write once.
#include "common.h"

// ---- im2col: x [n,H,W,C] -> col [n*Ho*Wo, Kp], column (kh*KW + kw)*C + c, zero outside the image and in the K padding ----
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int Kp, size_t rows) {
    const int kvec = Kp / 8, cvec = C / 8, K = KH * KW * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < rows * kvec; idx += (size_t)gridDim.x * 256) {
        const int kv = (int)(idx % kvec);
        const size_t row = idx / kvec;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (kv * 8 < K) {
            const int tap = kv / cvec, c = (kv - tap * cvec) * 8;
            const int kh = tap / KW, kw = tap - kh * KW;
            const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho);
            const size_t n = row / ((size_t)Wo * Ho);
            const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint4*>(x + ((n * H + iy) * W + ix) * C + c);
        }
        *reinterpret_cast<uint4*>(col + row * Kp + (size_t)kv * 8) = v;
    }
}

MMG_API int mmg_im2col_nhwc(const void* x, void* col, int n, int H, int W, int C, int KH, int KW, int stride, int pad, int Kp,
                            hipStream_t stream) {
    MMG_CHECK_ARG(x && col && n > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 &&
                      Kp >= KH * KW * C && Kp % 8 == 0, "mmg_im2col_nhwc: bad argument (C=%d must be a multiple of 8, Kp=%d)", C, Kp);
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    MMG_CHECK_ARG(Ho > 0 && Wo > 0, "mmg_im2col_nhwc: empty output");
    const size_t rows = (size_t)n * Ho * Wo, total = rows * (Kp / 8);
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)col, H, W, C, KH, KW, stride, pad,
                       Ho, Wo, Kp, rows);
    MMG_LAUNCH_CHECK("mmg_im2col_nhwc");
    return 0;
}

// ---- col2im (data gradient of a k x k convolution): dx[n,y,x,c] = sum over the taps that read this pixel; gather, no atomics ----
__global__ __launch_bounds__(256) void col2im_kernel(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dx, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int Kp, size_t pixels) {
    const int cvec = C / 8;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < pixels * cvec; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % cvec) * 8;
        const size_t pix = idx / cvec;
        const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
        const size_t n = pix / ((size_t)W * H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < KH; ++kh) {
            const int ty = iy + pad - kh;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int tx = ix + pad - kw;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(dcol + ((n * Ho + oy) * Wo + ox) * Kp + (size_t)(kh * KW + kw) * C + c);
                acc[0] += bf2f_lo(v.x); acc[1] += bf2f_hi(v.x); acc[2] += bf2f_lo(v.y); acc[3] += bf2f_hi(v.y);
                acc[4] += bf2f_lo(v.z); acc[5] += bf2f_hi(v.z); acc[6] += bf2f_lo(v.w); acc[7] += bf2f_hi(v.w);
            }
        }
        *reinterpret_cast<uint4*>(dx + pix * C + c) =
            make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
    }
}

MMG_API int mmg_col2im_nhwc(const void* dcol, void* dx, int n, int H, int W, int C, int KH, int KW, int stride, int pad, int Kp,
                            hipStream_t stream) {
    MMG_CHECK_ARG(dcol && dx && n > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 &&
                      Kp >= KH * KW * C && Kp % 8 == 0, "mmg_col2im_nhwc: bad argument");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const size_t pixels = (size_t)n * H * W, total = pixels * (C / 8);
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(col2im_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dcol, (bf16_t*)dx, H, W, C, KH, KW, stride, pad,
                       Ho, Wo, Kp, pixels);
    MMG_LAUNCH_CHECK("mmg_col2im_nhwc");
    return 0;
}

// ---- max pool 3x3, stride 2, padding 1 (forward only: everything below layer4 is frozen) --------------------------------------------
__global__ __launch_bounds__(256) void maxpool3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int H, int W, int C, int Ho,
                                                       int Wo, size_t opix) {
    const int cvec = C / 8;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < opix * cvec; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % cvec) * 8;
        const size_t pix = idx / cvec;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
        const size_t n = pix / ((size_t)Wo * Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if (ix < 0 || ix >= W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + ((n * H + iy) * W + ix) * C + c);
                const float f[8] = {bf2f_lo(v.x), bf2f_hi(v.x), bf2f_lo(v.y), bf2f_hi(v.y), bf2f_lo(v.z), bf2f_hi(v.z), bf2f_lo(v.w), bf2f_hi(v.w)};
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], f[e]);
            }
        }
        *reinterpret_cast<uint4*>(y + pix * C + c) = make_uint4(pack2bf(m[0], m[1]), pack2bf(m[2], m[3]), pack2bf(m[4], m[5]), pack2bf(m[6], m[7]));
    }
}

MMG_API int mmg_maxpool3x3s2_nhwc(const void* x, void* y, int n, int H, int W, int C, hipStream_t stream) {
    MMG_CHECK_ARG(x && y && n > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "mmg_maxpool3x3s2_nhwc: bad argument");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t opix = (size_t)n * Ho * Wo, total = opix * (C / 8);
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(maxpool3_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, H, W, C, Ho, Wo, opix);
    MMG_LAUNCH_CHECK("mmg_maxpool3x3s2_nhwc");
    return 0;
}

// ---- batch norm ----------------------------------------------------------------------------------------------------------------------
// column sums of x and x^2 over a slab of rows (fp32, accumulated with atomics; the caller zeroes sum / sumsq)
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16_t* __restrict__ x, int M, int C, int rows_per_block,
                                                       float* __restrict__ sum, float* __restrict__ sumsq) {
    const int cvec = C / 8;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
    // thread = (column group cv of 8 channels, row phase sub); 256 / cvec threads share a column group (cvec <= 256)
    const int cv = threadIdx.x % cvec, lanes = 256 / cvec, sub = threadIdx.x / cvec;
    if (sub >= lanes) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + sub; r < r1; r += lanes) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)r * C + cv * 8);
        const float f[8] = {bf2f_lo(v.x), bf2f_hi(v.x), bf2f_lo(v.y), bf2f_hi(v.y), bf2f_lo(v.z), bf2f_hi(v.z), bf2f_lo(v.w), bf2f_hi(v.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] = fmaf(f[e], f[e], q[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(sum + cv * 8 + e, s[e]); atomicAdd(sumsq + cv * 8 + e, q[e]); }
}

MMG_API int mmg_bn_stats(const void* x, int M, int C, float* sum, float* sumsq, hipStream_t stream) {
    MMG_CHECK_ARG(x && sum && sumsq && M > 0 && C >= 8 && C % 8 == 0 && C <= 2048, "mmg_bn_stats: M=%d C=%d (C multiple of 8, <= 2048)", M, C);
    int blocks = cdiv(M, 64);
    if (blocks > 1024) blocks = 1024;
    const int rpb = cdiv(M, blocks);
    blocks = cdiv(M, rpb);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, M, C, rpb, sum, sumsq);
    MMG_LAUNCH_CHECK("mmg_bn_stats");
    return 0;
}

// train != 0: mean / biased variance of the batch from (sum, sumsq, M); running statistics updated as torch does
// (running = (1 - momentum) running + momentum batch, unbiased variance).  train == 0: running statistics are used.
// Outputs: mean, rstd (for the backward) and the fused affine y = x * scale + shift.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, int M, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, int train, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float mu, var;
    if (train) {
        mu = sum[c] / M;
        var = fmaxf(sumsq[c] / M - mu * mu, 0.f);
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (M > 1 ? (float)M / (M - 1) : 1.f);
        }
    } else {
        mu = running_mean[c];
        var = running_var[c];
    }
    const float rs = rsqrtf(var + eps);
    mean[c] = mu; rstd[c] = rs;
    const float sc = gamma[c] * rs;
    scale[c] = sc; shift[c] = beta[c] - mu * sc;
}

MMG_API int mmg_bn_finalize(const float* sum, const float* sumsq, int M, int C, const float* gamma, const float* beta, float eps,
                            float momentum, float* running_mean, float* running_var, int train, float* mean, float* rstd,
                            float* scale, float* shift, hipStream_t stream) {
    MMG_CHECK_ARG(gamma && beta && mean && rstd && scale && shift && M > 0 && C > 0, "mmg_bn_finalize: bad argument");
    MMG_CHECK_ARG(train ? (sum && sumsq) : (running_mean && running_var), "mmg_bn_finalize: missing statistics for this mode");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, sum, sumsq, M, C, gamma, beta, eps, momentum,
                       running_mean, running_var, train, mean, rstd, scale, shift);
    MMG_LAUNCH_CHECK("mmg_bn_finalize");
    return 0;
}

// y = x * scale[c] + shift[c] (+ residual) (ReLU)
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const bf16_t* __restrict__ res,
                                                       bf16_t* __restrict__ y, int C, int relu, size_t nvec) {
    const int cvec = C / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cvec) * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(x + i * 8);
        float f[8] = {bf2f_lo(v.x), bf2f_hi(v.x), bf2f_lo(v.y), bf2f_hi(v.y), bf2f_lo(v.z), bf2f_hi(v.z), bf2f_lo(v.w), bf2f_hi(v.w)};
        uint4 rv = make_uint4(0, 0, 0, 0);
        if (res) rv = *reinterpret_cast<const uint4*>(res + i * 8);
        const float r[8] = {bf2f_lo(rv.x), bf2f_hi(rv.x), bf2f_lo(rv.y), bf2f_hi(rv.y), bf2f_lo(rv.z), bf2f_hi(rv.z), bf2f_lo(rv.w), bf2f_hi(rv.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f[e] = fmaf(f[e], scale[c + e], shift[c + e]) + r[e];
            if (relu) f[e] = fmaxf(f[e], 0.f);
        }
        *reinterpret_cast<uint4*>(y + i * 8) = make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
    }
}

MMG_API int mmg_bn_apply(const void* x, const float* scale, const float* shift, const void* residual, void* y, int M, int C, int relu,
                         hipStream_t stream) {
    MMG_CHECK_ARG(x && scale && shift && y && M > 0 && C >= 8 && C % 8 == 0, "mmg_bn_apply: bad argument");
    const size_t nvec = (size_t)M * C / 8;
    const int blocks = (int)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, scale, shift, (const bf16_t*)residual,
                       (bf16_t*)y, C, relu, nvec);
    MMG_LAUNCH_CHECK("mmg_bn_apply");
    return 0;
}

// Backward of y = relu?( bn(x) (+ residual) ) in training mode.  g = dy * (out > 0 when `out` is given: the ReLU mask from the
// layer's own output).  Pass 1 (reduce): sum_g[c] += g, sum_gx[c] += g * xhat.  Pass 2 (apply):
//   dx = gamma rstd (g - sum_g / M - xhat sum_gx / M);  dres (optional) = g  (the gradient of the residual branch).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ out, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int M, int C, int rows_per_block,
                                                            float* __restrict__ sum_g, float* __restrict__ sum_gx) {
    const int cvec = C / 8;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
    const int cv = threadIdx.x % cvec, lanes = 256 / cvec, sub = threadIdx.x / cvec;
    if (sub >= lanes) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = mean[cv * 8 + e]; rs[e] = rstd[cv * 8 + e]; }
    for (int r = r0 + sub; r < r1; r += lanes) {
        const size_t o = (size_t)r * C + cv * 8;
        const uint4 gv = *reinterpret_cast<const uint4*>(dy + o), xv = *reinterpret_cast<const uint4*>(x + o);
        uint4 ov = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        if (out) ov = *reinterpret_cast<const uint4*>(out + o);
        const float g[8] = {bf2f_lo(gv.x), bf2f_hi(gv.x), bf2f_lo(gv.y), bf2f_hi(gv.y), bf2f_lo(gv.z), bf2f_hi(gv.z), bf2f_lo(gv.w), bf2f_hi(gv.w)};
        const float xx[8] = {bf2f_lo(xv.x), bf2f_hi(xv.x), bf2f_lo(xv.y), bf2f_hi(xv.y), bf2f_lo(xv.z), bf2f_hi(xv.z), bf2f_lo(xv.w), bf2f_hi(xv.w)};
        const float oo[8] = {bf2f_lo(ov.x), bf2f_hi(ov.x), bf2f_lo(ov.y), bf2f_hi(ov.y), bf2f_lo(ov.z), bf2f_hi(ov.z), bf2f_lo(ov.w), bf2f_hi(ov.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ge = oo[e] > 0.f ? g[e] : 0.f;
            s[e] += ge; q[e] = fmaf(ge, (xx[e] - mu[e]) * rs[e], q[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(sum_g + cv * 8 + e, s[e]); atomicAdd(sum_gx + cv * 8 + e, q[e]); }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           const bf16_t* __restrict__ out, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sum_g, const float* __restrict__ sum_gx, int M,
                                                           int C, bf16_t* __restrict__ dx, bf16_t* __restrict__ dres, size_t nvec,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int cvec = C / 8;
    const float inv = 1.0f / M;
    if (blockIdx.x == 0 && dgamma) {                   // parameter gradients (accumulated): d gamma = sum g xhat, d beta = sum g
        for (int c = threadIdx.x; c < C; c += 256) { dgamma[c] += sum_gx[c]; dbeta[c] += sum_g[c]; }
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cvec) * 8;
        const uint4 gv = *reinterpret_cast<const uint4*>(dy + i * 8), xv = *reinterpret_cast<const uint4*>(x + i * 8);
        uint4 ov = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        if (out) ov = *reinterpret_cast<const uint4*>(out + i * 8);
        const float g[8] = {bf2f_lo(gv.x), bf2f_hi(gv.x), bf2f_lo(gv.y), bf2f_hi(gv.y), bf2f_lo(gv.z), bf2f_hi(gv.z), bf2f_lo(gv.w), bf2f_hi(gv.w)};
        const float xx[8] = {bf2f_lo(xv.x), bf2f_hi(xv.x), bf2f_lo(xv.y), bf2f_hi(xv.y), bf2f_lo(xv.z), bf2f_hi(xv.z), bf2f_lo(xv.w), bf2f_hi(xv.w)};
        const float oo[8] = {bf2f_lo(ov.x), bf2f_hi(ov.x), bf2f_lo(ov.y), bf2f_hi(ov.y), bf2f_lo(ov.z), bf2f_hi(ov.z), bf2f_lo(ov.w), bf2f_hi(ov.w)};
        float d[8], ge[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ge[e] = oo[e] > 0.f ? g[e] : 0.f;
            const float xh = (xx[e] - mean[c + e]) * rstd[c + e];
            d[e] = gamma[c + e] * rstd[c + e] * (ge[e] - sum_g[c + e] * inv - xh * sum_gx[c + e] * inv);
        }
        *reinterpret_cast<uint4*>(dx + i * 8) = make_uint4(pack2bf(d[0], d[1]), pack2bf(d[2], d[3]), pack2bf(d[4], d[5]), pack2bf(d[6], d[7]));
        if (dres) *reinterpret_cast<uint4*>(dres + i * 8) = make_uint4(pack2bf(ge[0], ge[1]), pack2bf(ge[2], ge[3]), pack2bf(ge[4], ge[5]), pack2bf(ge[6], ge[7]));
    }
}

MMG_API int mmg_bn_bwd_reduce(const void* dy, const void* x, const void* out, const float* mean, const float* rstd, int M, int C,
                              float* sum_g, float* sum_gx, hipStream_t stream) {
    MMG_CHECK_ARG(dy && x && mean && rstd && sum_g && sum_gx && M > 0 && C >= 8 && C % 8 == 0 && C <= 2048, "mmg_bn_bwd_reduce: bad argument");
    int blocks = cdiv(M, 64);
    if (blocks > 1024) blocks = 1024;
    const int rpb = cdiv(M, blocks);
    blocks = cdiv(M, rpb);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)out,
                       mean, rstd, M, C, rpb, sum_g, sum_gx);
    MMG_LAUNCH_CHECK("mmg_bn_bwd_reduce");
    return 0;
}

MMG_API int mmg_bn_bwd_apply(const void* dy, const void* x, const void* out, const float* mean, const float* rstd, const float* gamma,
                             const float* sum_g, const float* sum_gx, int M, int C, void* dx, void* dres, float* dgamma,
                             float* dbeta, hipStream_t stream) {
    MMG_CHECK_ARG(dy && x && mean && rstd && gamma && sum_g && sum_gx && dx && M > 0 && C >= 8 && C % 8 == 0, "mmg_bn_bwd_apply: bad argument");
    MMG_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "mmg_bn_bwd_apply: dgamma and dbeta go together");
    const size_t nvec = (size_t)M * C / 8;
    const int blocks = (int)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)out, mean,
                       rstd, gamma, sum_g, sum_gx, M, C, (bf16_t*)dx, (bf16_t*)dres, nvec, dgamma, dbeta);
    MMG_LAUNCH_CHECK("mmg_bn_bwd_apply");
    return 0;
}
