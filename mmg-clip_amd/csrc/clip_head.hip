// Contrastive head for mmg-clip on gfx950: L2-normalise, logits = s * X Y^T, symmetric
// cross-entropy and their backward, in fp32 on the f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// Reference arithmetic being replaced (citations into /root/reference):
//   mmgclip/networks/mmgclip_model.py:128-136  (e / ||e||, exp(logit_scale), two [n,n] matmuls)
//   mmgclip/loss/losses.py:36-44               (CLIPLoss: (CE(Li,arange)+CE(Lt,arange))/2)
//   mmgclip/loss/losses.py:74-91               (MMGCLIPLoss: same CE on recomputed logits + t2t term)
//
// Design: the [n_local, N] logit tile never has to reach HBM.  A workgroup owns 32 local rows i,
// keeps their embeddings in LDS and streams 32-column tiles of the gathered matrix Y.  The MFMA is
// issued "swapped" (A = Y tile, B = X tile) so each lane holds ONE row i (its column of the
// accumulator) and 16 columns j in registers: the row max / sum of the softmax are register-local
// and need a single lane^32 shuffle at the end.  In the backward the accumulator registers are fed
// back as the A operand of the second product (dX = G Y) with no lane movement.
#include "common.h"

#define CLIP_ROWS 32
#define CLIP_THREADS 256
#define NEG_BIG (-1.0e30f)

// row index inside a 32x32 f32 MFMA accumulator: reg r of lane-half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------
// L2 normalise rows (no epsilon, as the reference): y = x / ||x||_2
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ norm, int rows, int D) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { float v = xr[c]; s += v * v; }
    s = wave_sum(s);
    const float nrm = sqrtf(s);
    float* yr = y + (size_t)r * D;
    for (int c = lane; c < D; c += 64) yr[c] = xr[c] / nrm;
    if (lane == 0) norm[r] = nrm;
}

// dx = (dy - y * <y, dy>) / ||x||
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ norm,
                                                         const float* __restrict__ dy, float* __restrict__ dx,
                                                         int rows, int D) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* yr = y + (size_t)r * D;
    const float* gr = dy + (size_t)r * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += yr[c] * gr[c];
    s = wave_sum(s);
    const float inv = 1.0f / norm[r];
    float* dr = dx + (size_t)r * D;
    for (int c = lane; c < D; c += 64) dr[c] = (gr[c] - yr[c] * s) * inv;
}

// ---------------------------------------------------------------------------------------------
// Z^T tile: acc[reg] of lane (i = lane&31, h = lane>>5) = <Y[j0 + acc_row(reg,h)], X[i]>
// Xs: LDS image of the 32 local rows, row pitch xld floats.  yrow: this lane's (clamped) Y row.
// k order inside a step is arbitrary as long as A and B agree: half h takes k = 8c + 4h + e.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 zt_tile(const float* __restrict__ yrow, const float* Xs_row, int D, int h) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float4* yp = reinterpret_cast<const float4*>(yrow) + h;
    const float4* xp = reinterpret_cast<const float4*>(Xs_row) + h;
#pragma unroll 4
    for (int c = 0; c < D / 8; ++c) {
        const float4 yv = yp[2 * c];
        const float4 xv = xp[2 * c];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yv.x, xv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yv.y, xv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yv.z, xv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yv.w, xv.w, acc, 0, 0, 0);
    }
    return acc;
}

// Stage the workgroup's 32 rows of X into LDS (rows beyond n_loc are clamped duplicates).
__device__ __forceinline__ void stage_x(const float* __restrict__ X, float* Xs, int i0, int n_loc, int D, int xld) {
    const int nvec = D / 4;
    for (int idx = threadIdx.x; idx < CLIP_ROWS * nvec; idx += CLIP_THREADS) {
        const int r = idx / nvec, c = idx - r * nvec;
        const int gr = min(i0 + r, n_loc - 1);
        const float4 v = reinterpret_cast<const float4*>(X + (size_t)gr * D)[c];
        *reinterpret_cast<float4*>(Xs + r * xld + 4 * c) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Forward: per local row i, lse[i] = logsumexp_j(s <X_i, Y_j>) and pos[i] = s <X_i, Y_{label(i)}>.
// label(i) = diag_off + i (the positive of local row i in the gathered matrix).  Optionally writes the logits.
// ---------------------------------------------------------------------------------------------
template <bool WRITE_LOGITS>
__global__ __launch_bounds__(CLIP_THREADS) void clip_rows_fwd_kernel(
    const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ scale_ptr, int n_loc, int N,
    int D, int diag_off, float* __restrict__ lse, float* __restrict__ pos, float* __restrict__ logits, int ldl) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int xld = D + 4;
    float* Xs = smem;                       // [32][xld]
    float* red = smem + CLIP_ROWS * xld;    // [4 waves][32 rows][3] (m, l, pos)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int il = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * CLIP_ROWS;
    const int ig = i0 + il;
    const float scale = *scale_ptr;

    stage_x(X, Xs, i0, n_loc, D, xld);
    __syncthreads();

    float m = NEG_BIG, l = 0.f, p = 0.f;
    const int lab = diag_off + ig;
    const int ntiles = (N + 31) / 32;
    for (int jt = wave; jt < ntiles; jt += 4) {
        const int j0 = jt * 32;
        const float* yrow = Y + (size_t)min(j0 + il, N - 1) * D;
        const f32x16 acc = zt_tile(yrow, Xs + il * xld, D, h);
        float z[16];
        float tmax = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + acc_row(r, h);
            z[r] = acc[r] * scale;
            if (j < N) {
                tmax = fmaxf(tmax, z[r]);
                if (j == lab) p = z[r];
            }
        }
        const float mn = fmaxf(m, tmax);
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + acc_row(r, h);
            if (j < N) s += __expf(z[r] - mn);
        }
        l = l * __expf(m - mn) + s;
        m = mn;
        if (WRITE_LOGITS && ig < n_loc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + 8 * q + 4 * h;
                float* dst = logits + (size_t)ig * ldl + j;
                if (j + 3 < N && ((ldl & 3) == 0)) {
                    *reinterpret_cast<float4*>(dst) = make_float4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j + e < N) dst[e] = z[4 * q + e];
                }
            }
        }
    }
    // combine the two lane halves (same row i, disjoint columns)
    {
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64), p2 = __shfl_xor(p, 32, 64);
        const float mn = fmaxf(m, m2);
        l = l * __expf(m - mn) + l2 * __expf(m2 - mn);
        m = mn;
        p += p2;
    }
    if (h == 0) {
        red[(wave * 32 + il) * 3 + 0] = m;
        red[(wave * 32 + il) * 3 + 1] = l;
        red[(wave * 32 + il) * 3 + 2] = p;
    }
    __syncthreads();
    if (threadIdx.x < 32 && i0 + threadIdx.x < n_loc) {
        float mm = NEG_BIG, ll = 0.f, pp = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float m2 = red[(w * 32 + threadIdx.x) * 3 + 0], l2 = red[(w * 32 + threadIdx.x) * 3 + 1];
            const float mn = fmaxf(mm, m2);
            ll = ll * __expf(mm - mn) + l2 * __expf(m2 - mn);
            mm = mn;
            pp += red[(w * 32 + threadIdx.x) * 3 + 2];
        }
        lse[i0 + threadIdx.x] = mm + __logf(ll);
        pos[i0 + threadIdx.x] = pp;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward of the rows: dX[i,:] = s * sum_j g_ij Y[j,:],  dscale += sum_ij g_ij <X_i,Y_j>
//   FUSED : g_ij = coef * (exp(z_ij - lse_row[i]) + exp(z_ij - lse_col[j]) - 2 [j == diag_off+i])
//           (the gradient of the symmetric CE w.r.t. z_ij when lse_row / lse_col are the row and the
//            column log-sum-exps of the GLOBAL logit matrix; coef = dloss / (2 N_global))
//   DENSE : g_ij = Ga[i*lda + j] + Gb[j*ldb + i]   (upstream gradients of materialised logits and of
//            their transposed twin, either may be null)
// NDT = 32-wide d tiles per wave (D <= 128*NDT).
// ---------------------------------------------------------------------------------------------
template <bool FUSED, int NDT>
__global__ __launch_bounds__(CLIP_THREADS) void clip_rows_bwd_kernel(
    const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ scale_ptr,
    const float* __restrict__ lse_row, const float* __restrict__ lse_col, const float* __restrict__ gout, float coef,
    const float* __restrict__ Ga, int lda, const float* __restrict__ Gb, int ldb, int n_loc, int N, int D,
    int diag_off, float* __restrict__ dX, float* __restrict__ dscale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int xld = D + 4;
    float* Xs = smem;                      // [32][xld]
    float* Gs = smem + CLIP_ROWS * xld;    // [4 slots][32 j][33]
    float* red = Gs + 4 * 32 * 33;         // [4]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int il = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * CLIP_ROWS;
    const int ig = i0 + il;
    const float scale = *scale_ptr;
    const float c_up = FUSED ? coef * (gout ? *gout : 1.0f) : 1.0f;

    stage_x(X, Xs, i0, n_loc, D, xld);
    __syncthreads();

    f32x16 dacc[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[t][r] = 0.f;

    const float lr = (FUSED && ig < n_loc) ? lse_row[ig] : 0.f;
    const int lab = diag_off + ig;
    float ds = 0.f;
    const int ntiles = (N + 31) / 32;
    const int ndtiles = D / 32;
    for (int jt0 = 0; jt0 < ntiles; jt0 += 4) {
        const int jt = jt0 + wave;
        float* Gw = Gs + wave * 32 * 33;
        if (jt < ntiles) {
            const int j0 = jt * 32;
            const float* yrow = Y + (size_t)min(j0 + il, N - 1) * D;
            const f32x16 acc = zt_tile(yrow, Xs + il * xld, D, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = acc_row(r, h);
                const int j = j0 + jl;
                float g = 0.f;
                if (j < N && ig < n_loc) {
                    if (FUSED) {
                        const float z = acc[r] * scale;
                        g = __expf(z - lr) + __expf(z - lse_col[j]);
                        if (j == lab) g -= 2.0f;
                        g *= c_up;
                    } else {
                        if (Ga) g += Ga[(size_t)ig * lda + j];
                        if (Gb) g += Gb[(size_t)j * ldb + ig];
                    }
                    ds += g * acc[r];
                }
                Gw[jl * 33 + il] = g;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) Gw[acc_row(r, h) * 33 + il] = 0.f;
        }
        __syncthreads();
        // second product: every wave takes its d tiles over the 4 freshly written G tiles
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int j0 = (jt0 + s) * 32;
            if (j0 < N) {
                const float* Gsl = Gs + s * 32 * 33;
#pragma unroll
                for (int t = 0; t < NDT; ++t) {
                    const int dt = wave + 4 * t;
                    if (dt < ndtiles) {
                        const float* ycol = Y + dt * 32 + il;
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) {
                            const int jl = 2 * kk + h;
                            const float a = Gsl[jl * 33 + il];                             // A[i][k=jl]
                            const float b = ycol[(size_t)min(j0 + jl, N - 1) * D];          // B[k=jl][d]
                            dacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, dacc[t], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // dX tile: lane holds column d = dt*32 + il, rows acc_row(r,h)
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
        const int dt = wave + 4 * t;
        if (dt < ndtiles) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + acc_row(r, h);
                if (i < n_loc) dX[(size_t)i * D + dt * 32 + il] = dacc[t][r] * scale;
            }
        }
    }
    if (dscale) {
        ds = wave_sum(ds);
        if (lane == 0) red[wave] = ds;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(dscale, red[0] + red[1] + red[2] + red[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// Cross-entropy over materialised logits (CLIPLoss / AveragedMedicalCLIPLoss operands):
// one wave per row; loss_sum += weight * (lse - z[label]).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_rows_fwd_kernel(const float* __restrict__ logits, int ld,
                                                          const long long* __restrict__ labels, int rows, int C,
                                                          float weight, float* __restrict__ lse,
                                                          float* __restrict__ loss_sum) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* zr = logits + (size_t)r * ld;
    float m = NEG_BIG;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, zr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(zr[c] - m);
    s = wave_sum(s);
    if (lane == 0) {
        const float l = m + __logf(s);
        lse[r] = l;
        const long long lab = labels ? labels[r] : (long long)r;
        atomicAdd(loss_sum, weight * (l - zr[lab]));
    }
}

// dlogits[r,c] = gout * weight * (exp(z - lse[r]) - [c == label])
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const float* __restrict__ logits, int ld,
                                                          const long long* __restrict__ labels,
                                                          const float* __restrict__ lse, const float* __restrict__ gout,
                                                          float weight, int rows, int C, float* __restrict__ dlogits,
                                                          int ldd) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float g = weight * (gout ? *gout : 1.0f);
    const float l = lse[r];
    const long long lab = labels ? labels[r] : (long long)r;
    const float* zr = logits + (size_t)r * ld;
    float* dr = dlogits + (size_t)r * ldd;
    for (int c = lane; c < C; c += 64) dr[c] = g * (__expf(zr[c] - l) - (c == lab ? 1.0f : 0.0f));
}

// loss = coef * (sum(lse_a - pos_a) + sum(lse_b - pos_b)) accumulated into *loss (one block)
__global__ __launch_bounds__(256) void clip_loss_reduce_kernel(const float* __restrict__ lse_a,
                                                               const float* __restrict__ pos_a,
                                                               const float* __restrict__ lse_b,
                                                               const float* __restrict__ pos_b, int n, float coef,
                                                               float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        s += lse_a[i] - pos_a[i];
        if (lse_b) s += lse_b[i] - pos_b[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, coef * (red[0] + red[1] + red[2] + red[3]));
}

// =============================================================================================
// C ABI
// =============================================================================================
MMG_API int mmg_l2norm_fwd(const float* x, float* y, float* norm, int rows, int D, hipStream_t stream) {
    MMG_CHECK_ARG(x && y && norm && rows > 0 && D > 0, "mmg_l2norm_fwd: bad argument");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, y, norm, rows, D);
    MMG_LAUNCH_CHECK("mmg_l2norm_fwd");
    return 0;
}

MMG_API int mmg_l2norm_bwd(const float* y, const float* norm, const float* dy, float* dx, int rows, int D,
                           hipStream_t stream) {
    MMG_CHECK_ARG(y && norm && dy && dx && rows > 0 && D > 0, "mmg_l2norm_bwd: bad argument");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, y, norm, dy, dx, rows, D);
    MMG_LAUNCH_CHECK("mmg_l2norm_bwd");
    return 0;
}

static int clip_check_dims(const char* who, int n_loc, int N, int D) {
    MMG_CHECK_ARG(n_loc > 0 && N > 0, "%s: n_loc=%d N=%d must be positive", who, n_loc, N);
    MMG_CHECK_ARG(D >= 32 && D % 32 == 0 && D <= 1024, "%s: D=%d must be a multiple of 32 in [32,1024]", who, D);
    return 0;
}

MMG_API int mmg_clip_rows_fwd(const float* X, const float* Y, const float* scale, int n_loc, int N, int D,
                              int diag_off, float* lse, float* pos, float* logits, int ldl, hipStream_t stream) {
    if (clip_check_dims("mmg_clip_rows_fwd", n_loc, N, D)) return 1;
    MMG_CHECK_ARG(X && Y && scale && lse && pos, "mmg_clip_rows_fwd: null pointer");
    MMG_CHECK_ARG(!logits || ldl >= N, "mmg_clip_rows_fwd: ldl=%d < N=%d", ldl, N);
    const size_t shm = (size_t)(CLIP_ROWS * (D + 4) + 4 * 32 * 3) * sizeof(float);
    const dim3 grid(cdiv(n_loc, CLIP_ROWS));
    if (logits) {
        mmg_allow_lds(clip_rows_fwd_kernel<true>, shm);
        hipLaunchKernelGGL(clip_rows_fwd_kernel<true>, grid, dim3(CLIP_THREADS), shm, stream, X, Y, scale, n_loc, N, D,
                           diag_off, lse, pos, logits, ldl);
    } else {
        mmg_allow_lds(clip_rows_fwd_kernel<false>, shm);
        hipLaunchKernelGGL(clip_rows_fwd_kernel<false>, grid, dim3(CLIP_THREADS), shm, stream, X, Y, scale, n_loc, N, D,
                           diag_off, lse, pos, (float*)nullptr, 0);
    }
    MMG_LAUNCH_CHECK("mmg_clip_rows_fwd");
    return 0;
}

template <bool FUSED>
static int launch_rows_bwd(const float* X, const float* Y, const float* scale, const float* lse_row,
                           const float* lse_col, const float* gout, float coef, const float* Ga, int lda,
                           const float* Gb, int ldb, int n_loc, int N, int D, int diag_off, float* dX, float* dscale,
                           hipStream_t stream) {
    const size_t shm = (size_t)(CLIP_ROWS * (D + 4) + 4 * 32 * 33 + 4) * sizeof(float);
    const dim3 grid(cdiv(n_loc, CLIP_ROWS));
    const int ndt = (D + 127) / 128;
#define LAUNCH_BWD(NDT)                                                                                               \
    do {                                                                                                              \
        mmg_allow_lds(clip_rows_bwd_kernel<FUSED, NDT>, shm);                                                         \
        hipLaunchKernelGGL((clip_rows_bwd_kernel<FUSED, NDT>), grid, dim3(CLIP_THREADS), shm, stream, X, Y, scale,    \
                           lse_row, lse_col, gout, coef, Ga, lda, Gb, ldb, n_loc, N, D, diag_off, dX, dscale);        \
    } while (0)
    if (ndt <= 1) LAUNCH_BWD(1);
    else if (ndt == 2) LAUNCH_BWD(2);
    else if (ndt <= 4) LAUNCH_BWD(4);
    else LAUNCH_BWD(8);
#undef LAUNCH_BWD
    return 0;
}

// Fused symmetric-CE gradient of the rows (see kernel comment).  dscale (may be null) is ACCUMULATED.
MMG_API int mmg_clip_rows_bwd_fused(const float* X, const float* Y, const float* scale, const float* lse_row,
                                    const float* lse_col, const float* gout, float coef, int n_loc, int N, int D,
                                    int diag_off, float* dX, float* dscale, hipStream_t stream) {
    if (clip_check_dims("mmg_clip_rows_bwd_fused", n_loc, N, D)) return 1;
    MMG_CHECK_ARG(X && Y && scale && lse_row && lse_col && dX, "mmg_clip_rows_bwd_fused: null pointer");
    launch_rows_bwd<true>(X, Y, scale, lse_row, lse_col, gout, coef, nullptr, 0, nullptr, 0, n_loc, N, D, diag_off, dX,
                          dscale, stream);
    MMG_LAUNCH_CHECK("mmg_clip_rows_bwd_fused");
    return 0;
}

// Backward of materialised logits: dX = s * (Ga + Gb^T) Y ; dscale += <Ga + Gb^T, X Y^T>.
MMG_API int mmg_clip_rows_bwd_dense(const float* X, const float* Y, const float* scale, const float* Ga, int lda,
                                    const float* Gb, int ldb, int n_loc, int N, int D, float* dX, float* dscale,
                                    hipStream_t stream) {
    if (clip_check_dims("mmg_clip_rows_bwd_dense", n_loc, N, D)) return 1;
    MMG_CHECK_ARG(X && Y && scale && dX && (Ga || Gb), "mmg_clip_rows_bwd_dense: null pointer");
    MMG_CHECK_ARG((!Ga || lda >= N) && (!Gb || ldb >= n_loc), "mmg_clip_rows_bwd_dense: bad leading dimension");
    launch_rows_bwd<false>(X, Y, scale, nullptr, nullptr, nullptr, 1.0f, Ga, lda, Gb, ldb, n_loc, N, D, 0, dX, dscale,
                           stream);
    MMG_LAUNCH_CHECK("mmg_clip_rows_bwd_dense");
    return 0;
}

MMG_API int mmg_ce_rows_fwd(const float* logits, int ld, const long long* labels, int rows, int C, float weight,
                            float* lse, float* loss_sum, hipStream_t stream) {
    MMG_CHECK_ARG(logits && lse && loss_sum && rows > 0 && C > 0 && ld >= C, "mmg_ce_rows_fwd: bad argument");
    hipLaunchKernelGGL(ce_rows_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, logits, ld, labels, rows, C,
                       weight, lse, loss_sum);
    MMG_LAUNCH_CHECK("mmg_ce_rows_fwd");
    return 0;
}

MMG_API int mmg_ce_rows_bwd(const float* logits, int ld, const long long* labels, const float* lse, const float* gout,
                            float weight, int rows, int C, float* dlogits, int ldd, hipStream_t stream) {
    MMG_CHECK_ARG(logits && lse && dlogits && rows > 0 && C > 0 && ld >= C && ldd >= C, "mmg_ce_rows_bwd: bad argument");
    hipLaunchKernelGGL(ce_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, logits, ld, labels, lse, gout,
                       weight, rows, C, dlogits, ldd);
    MMG_LAUNCH_CHECK("mmg_ce_rows_bwd");
    return 0;
}

MMG_API int mmg_clip_loss_reduce(const float* lse_a, const float* pos_a, const float* lse_b, const float* pos_b, int n,
                                 float coef, float* loss, hipStream_t stream) {
    MMG_CHECK_ARG(lse_a && pos_a && loss && n > 0 && (!lse_b || pos_b), "mmg_clip_loss_reduce: bad argument");
    hipLaunchKernelGGL(clip_loss_reduce_kernel, dim3(1), dim3(256), 0, stream, lse_a, pos_a, lse_b, pos_b, n, coef, loss);
    MMG_LAUNCH_CHECK("mmg_clip_loss_reduce");
    return 0;
}
