/*
 * mmgclip_hip.h — C ABI of libmmgclip_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels behind
 * mmg-clip's contrastive-training hot path.
 *
 * The reference (abdel-habib/mmg-clip) has no FFI of its own: its hot path is Python calling ATen.  Each entry
 * below therefore cites the reference Python call site (path:line in the reference tree) whose arithmetic it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocations in this repo); nothing is
 *     allocated, freed or retained by the library; no global state besides a thread-local error string;
 *   - `stream` is a hipStream_t passed as void* (PyTorch's current stream); every call is asynchronous on it
 *     and safe to capture into a hipGraph;
 *   - matrices are row-major; `ld*` are leading dimensions in ELEMENTS;
 *   - bf16 tensors are passed as `void*` (raw 16-bit payload), fp32 as `float*`, token ids as `long long*`;
 *   - return value: 0 = launched, non-zero = rejected (bad shape / null pointer / launch error) and
 *     mmg_last_error() holds the reason.  Host wrappers raise on non-zero.
 *
 * This header is the single source of truth: mmgclip/_hip.py parses it to build the ctypes prototypes and
 * tests/test_abi.py checks that the library exports every symbol declared here.
 */
#ifndef MMGCLIP_HIP_H
#define MMGCLIP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mmg_stream_t; /* hipStream_t */

/* ---- library probe ------------------------------------------------------------------------------------ */
int mmg_abi_version(void);
const char* mmg_last_error(void);
const char* mmg_target_arch(void);
int mmg_device_cu_count(void);

/* ---- contrastive head (fp32, f32-input MFMA) ------------------------------------------------------------ */

/* y = x / ||x||_2 per row, no epsilon; norm[r] = ||x_r||.
 * Replaces mmgclip/networks/mmgclip_model.py:128-129. */
int mmg_l2norm_fwd(const float* x, float* y, float* norm, int rows, int D, mmg_stream_t stream);
/* dx = (dy - y <y,dy>) / norm   (autograd of the line above) */
int mmg_l2norm_bwd(const float* y, const float* norm, const float* dy, float* dx, int rows, int D,
                   mmg_stream_t stream);

/* For the n_loc local rows X against the N gathered rows Y (both L2-normalised, [.,D] fp32):
 *   lse[i] = logsumexp_j( s <X_i,Y_j> ),  pos[i] = s <X_i, Y_{diag_off+i}>,  s = *scale
 * and, when logits != NULL, logits[i*ldl + j] = s <X_i,Y_j>.
 * Replaces mmgclip/networks/mmgclip_model.py:132-136 (logit_scale * I @ T.t()) fused with the
 * log-softmax of mmgclip/loss/losses.py:40-41.  D % 32 == 0, D <= 1024. */
int mmg_clip_rows_fwd(const float* X, const float* Y, const float* scale, int n_loc, int N, int D, int diag_off,
                      float* lse, float* pos, float* logits, int ldl, mmg_stream_t stream);

/* Gradient of the symmetric cross-entropy w.r.t. the local rows, without materialising logits:
 *   g_ij = coef * gout * ( exp(z_ij - lse_row[i]) + exp(z_ij - lse_col[j]) - 2 [j == diag_off+i] )
 *   dX[i,:] = s * sum_j g_ij Y[j,:]        dscale += sum_ij g_ij <X_i,Y_j>   (dscale may be NULL)
 * lse_row: [n_loc] row log-sum-exps of this block; lse_col: [N] log-sum-exps of the transposed problem
 * (all ranks').  gout: device scalar (NULL = 1).  coef = 1/(2 N_global).
 * Autograd of mmgclip/loss/losses.py:39-43 through mmgclip/networks/mmgclip_model.py:135-136. */
int mmg_clip_rows_bwd_fused(const float* X, const float* Y, const float* scale, const float* lse_row,
                            const float* lse_col, const float* gout, float coef, int n_loc, int N, int D,
                            int diag_off, float* dX, float* dscale, mmg_stream_t stream);

/* Backward of materialised logits L = s X Y^T (and of the twin L' = s Y X^T):
 *   dX = s (Ga + Gb^T) Y,   dscale += <Ga + Gb^T, X Y^T>;   Ga: [n_loc,N] (ld lda), Gb: [N,n_loc] (ld ldb),
 * either may be NULL.  Autograd of mmgclip/networks/mmgclip_model.py:135-136 for arbitrary consumers. */
int mmg_clip_rows_bwd_dense(const float* X, const float* Y, const float* scale, const float* Ga, int lda,
                            const float* Gb, int ldb, int n_loc, int N, int D, float* dX, float* dscale,
                            mmg_stream_t stream);

/* Cross-entropy over materialised logits [rows,C]: lse[r] and loss_sum += weight * (lse[r] - z[r,label_r]);
 * labels NULL = arange.  Replaces F.cross_entropy at mmgclip/loss/losses.py:40-41,79-80,88-89,209-210. */
int mmg_ce_rows_fwd(const float* logits, int ld, const long long* labels, int rows, int C, float weight,
                    float* lse, float* loss_sum, mmg_stream_t stream);
/* dlogits = gout * weight * (softmax(z) - onehot(label)) */
int mmg_ce_rows_bwd(const float* logits, int ld, const long long* labels, const float* lse, const float* gout,
                    float weight, int rows, int C, float* dlogits, int ldd, mmg_stream_t stream);

/* loss += coef * ( sum_i (lse_a[i]-pos_a[i]) + sum_i (lse_b[i]-pos_b[i]) );  lse_b/pos_b may be NULL. */
int mmg_clip_loss_reduce(const float* lse_a, const float* pos_a, const float* lse_b, const float* pos_b, int n,
                         float coef, float* loss, mmg_stream_t stream);

/* ---- bf16 MFMA GEMMs (encoder towers, projection heads) -------------------------------------------------- */

/* C[M,N] = epilogue( alpha * A[M,K] B[N,K]^T + bias )   A,B bf16 row-major (K contiguous), fp32 accumulate.
 * epilogue (in this order): + bias[N];  activation by `epi`:
 *     0 none | 1 GELU(erf) (aux_out, when given, receives the pre-activation) | 2 multiply by GELU'(aux_in)
 *     3 ReLU (aux_out as 1)  | 4 ReLU' (gate by aux_in > 0);
 * then * colscale[N] (ConvNeXt layer scale), + residual[M,N] (bf16); C is bf16 (out_f32 = 0) or fp32.
 * K % 32 == 0, N % 8 == 0, leading dimensions multiples of 8.
 * Replaces nn.Linear forward / data-gradient in HF BertLayer (reference call site mmgclip/networks/encoder.py:156),
 * torchvision CNBlock + patchify convs (mmgclip/networks/encoder.py:53, mmgclip/networks/image_features.py:100)
 * and the projection heads (mmgclip/networks/projection.py:33,55-61,94-101). */
int mmg_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const float* bias, const float* colscale, const void* residual, int ldr, const void* aux_in,
                     int ldai, void* aux_out, int ldao, int epi, int out_f32, float alpha, mmg_stream_t stream);

/* Weight gradient: C[N1,N2] += alpha * A[M,N1]^T B[M,N2]  (A,B bf16; C fp32, accumulated with atomics, so the
 * caller zeroes C once per optimisation step).  Autograd of the linears above w.r.t. their weights. */
int mmg_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                     float alpha, mmg_stream_t stream);

/* Bias gradient: out[n] += sum_m A[m,n]  (A bf16 [M,N]). */
int mmg_colsum_bf16(const void* A, int lda, int M, int N, float* out, mmg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMGCLIP_HIP_H */
