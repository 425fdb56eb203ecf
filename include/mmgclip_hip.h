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
 *     allocated, freed or retained by the library; no global state besides a thread-local error string (and the opt-in
 *     kernel-name note of the diagnostics below);
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
/* Diagnostics for the measurement leg (bench.py `roofline`, mmgclip/profile.py): while notes are on, every dispatcher records the
 * name of the kernel instantiation it launches - spelled as rocprofv3's kernel trace spells it, e.g.
 * "gemm_nt_kernel<256, 256, 64, 4, 2, 0>" - and mmg_last_kernel() returns the calling thread's last one ("" when none).  No
 * reference counterpart: the reference has no kernels of its own to name. */
int mmg_set_kernel_notes(int on);
const char* mmg_last_kernel(void);

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
 *     (aux_out, when given, receives GELU(aux_in): the activation rebuilt for the weight-gradient GEMM)
 *     3 ReLU (aux_out as 1)  | 4 ReLU' (gate by aux_in > 0; aux_out as 2)  | 5 multiply by GELU'(aux_in) only (ABI 4: for callers that
 *     kept GELU(h) from the forward; aux_out unused) | 6 GELU(erf) whose aux_out receives GELU'(pre-activation) instead of the pre-activation
 *     (ABI 5) | 7 multiply by aux_in (ABI 5: the data gradient of a layer whose forward ran 6 / kept GELU');
 *     bf16 / e4m3 outputs of 1, 2, 5 and 6 evaluate GELU / GELU' by the polynomials of csrc/common.h (6e-4 relative), fp32 outputs by the erf form (1.5e-7);
 * then * colscale[N] (ConvNeXt layer scale), + residual[M,N] (bf16); C is bf16 (out_f32 = 0) or fp32.
 * K % 32 == 0, N % 8 == 0, leading dimensions multiples of 8.
 * Replaces nn.Linear forward / data-gradient in HF BertLayer (reference call site mmgclip/networks/encoder.py:156),
 * torchvision CNBlock + patchify convs (mmgclip/networks/encoder.py:53, mmgclip/networks/image_features.py:100)
 * and the projection heads (mmgclip/networks/projection.py:33,55-61,94-101). */
int mmg_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const float* bias, const float* colscale, const void* residual, int ldr, const void* aux_in,
                     int ldai, void* aux_out, int ldao, int epi, int out_f32, float alpha, mmg_stream_t stream);

/* Weight gradient: C[N1,N2] += alpha * A[M,N1]^T B[M,N2]  (A,B bf16; C fp32, accumulated with atomics, so the
 * caller zeroes C once per optimisation step).  colsum_a (nullable, fp32 [N1]) += alpha * column sums of A: the bias
 * gradient of the same linear, fused as one extra MFMA per fragment.  Autograd of the linears above. */
int mmg_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                     float alpha, float* colsum_a, mmg_stream_t stream);

/* Bias gradient: out[n] += sum_m A[m,n]  (A bf16 [M,N]). */
int mmg_colsum_bf16(const void* A, int lda, int M, int N, float* out, mmg_stream_t stream);

/* ---- LayerNorm / elementwise / pooling / optimiser (HBM-bound) --------------------------------------------- */

/* y = (x - mean) * rstd * gamma + beta over the last dim C of bf16 x[M,C]; mean/rstd (fp32 [M], nullable) are saved
 * for the backward.  patch != 0: row m = (n,h,w) of an [n,H,W] grid is written to the 2x2-patchified position
 * row (n,h/2,w/2), columns ((h&1)*2+(w&1))*C of a [n*(H/2)*(W/2), 4C] matrix (ConvNeXt downsample: the following
 * 2x2/s2 convolution becomes a GEMM; an odd last row / column is dropped, as that convolution does).  Replaces nn.LayerNorm / LayerNorm2d of torchvision ConvNeXt and HF BERT
 * (mmgclip/networks/encoder.py:53,156) and of MLPProjectionHead (mmgclip/networks/projection.py:100). */
int mmg_layernorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, float eps, void* y, int ldy,
                      float* mean, float* rstd, int M, int C, int patch, int H, int W, mmg_stream_t stream);
/* The same LayerNorm on an fp32 input row (HF BertLayer's LayerNorm(x + sublayer(x)), mmgclip/networks/encoder.py:156, with the
 * sum kept in fp32).  res (fp32 [M,C], nullable): x <- x + res IN PLACE first (x then holds the pre-LayerNorm sum the backward
 * needs).  y bf16 (operand of the next GEMM) and, when yf != NULL, an fp32 copy of it (the residual of the next sum). */
int mmg_layernorm_fwd_f32(float* x, int ldx, const float* res, int ldres, const float* gamma, const float* beta, float eps,
                          void* y, int ldy, float* yf, int ldyf, float* mean, float* rstd, int M, int C, mmg_stream_t stream);
int mmg_layernorm_bwd_f32(const void* dy, int lddy, const float* x, int ldx, const float* mean, const float* rstd,
                          const float* gamma, void* dx, int lddx, float* dgamma, float* dbeta, int M, int C,
                          const void* add, int ldadd, mmg_stream_t stream);
/* dx (bf16) and dgamma/dbeta (fp32, ACCUMULATED; both NULL to skip) given dy in the layout the forward wrote;
 * add (bf16 [M,C], nullable) is added to dx: the residual-path gradient of pre-LN transformer blocks. */
int mmg_layernorm_bwd(const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                      const float* gamma, void* dx, int lddx, float* dgamma, float* dbeta, int M, int C, int patch,
                      int H, int W, const void* add, int ldadd, mmg_stream_t stream);

/* y = GELU(x) elementwise, bf16, n % 8 == 0 (rebuilds the FFN activation in the backward pass). */
int mmg_gelu_fwd_bf16(const void* x, void* y, long long n, mmg_stream_t stream);
/* out = dy * act'(pre) elementwise, bf16; kind 0 = GELU(erf), 1 = ReLU, 2 = GELU(erf) by the exp-free polynomial of mmg_cnblock_bwdw; n % 8 == 0. */
int mmg_act_grad_bf16(const void* dy, const void* pre, void* out, long long n, int kind, mmg_stream_t stream);
/* dtype conversions of flat buffers */
int mmg_cast_f32_bf16(const float* x, void* y, long long n, mmg_stream_t stream);
int mmg_cast_bf16_f32(const void* x, float* y, long long n, mmg_stream_t stream);
/* dst[c,r] (bf16, ld ldd) = rowscale[r] * src[r,c] (fp32 [R,C]); rowscale nullable.  Transposed working copy of a
 * weight for the data-gradient GEMM (layer scale folded in for ConvNeXt's second linear). */
int mmg_transpose_cast_bf16(const float* src, int R, int C, const float* rowscale, void* dst, int ldd,
                            mmg_stream_t stream);

/* y[n,:] = mean over HW rows of bf16 x[n,HW,C] (fp32 out).  AdaptiveAvgPool2d(1), mmgclip/networks/encoder.py:54. */
int mmg_avgpool_fwd(const void* x, float* y, int n, int HW, int C, mmg_stream_t stream);
int mmg_avgpool_bwd(const float* dy, void* dx, int n, int HW, int C, mmg_stream_t stream);

/* Stem im2col: fp32 pixels [n,Cin,H,W] -> bf16 rows (n,h/P,w/P) x (kh,kw,cin), zero-padded to Kp columns;
 * scale16 != 0 applies ((65535 x) - 32767.5)/32767.5 (mmgclip/networks/image_features.py:95-99).  H, W need not be
 * multiples of P: the remainder rows / columns are ignored, as a stride-P convolution does. */
int mmg_patchify(const float* img, void* out, int n, int Cin, int H, int W, int P, int Kp, int scale16,
                 mmg_stream_t stream);

/* torch.optim.AdamW step over a flat fp32 buffer (mmgclip/experiments/ClassifierExperiment.py:74,118);
 * p_bf16 (nullable) receives the refreshed bf16 working copy.  step counts from 1. */
int mmg_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, mmg_stream_t stream);

/* ConvNeXt layer scale (out = x + gamma * (G W2^T + b2)): converts the unscaled weight-gradient GEMM results
 * dW2raw [C,K], db2raw [C] into dgamma, dW2, db2 (all ACCUMULATED).  torchvision CNBlock.layer_scale. */
int mmg_layerscale_finalize(const float* W2, const float* b2, const float* gamma, const float* dW2raw,
                            const float* db2raw, float* dW2, float* db2, float* dgamma, int C, int K,
                            mmg_stream_t stream);
/* dst += relayout(src): mode 0 = [R,(kh,kw,ci)] -> [R,CI,KH,KW] (patchify-conv weight gradients),
 * mode 1 = [49,R] -> [R,49] (depthwise taps). */
int mmg_grad_relayout(const float* src, float* dst, int mode, int R, int CI, int KH, int KW, int ld_src,
                      mmg_stream_t stream);

/* Inverted dropout on fp32 (projection heads: mmgclip/networks/projection.py:50,59,91,98); keep = uint8 mask. */
int mmg_dropout_fwd(const float* x, float* y, void* keep, long long n, float p, long long seed, mmg_stream_t stream);
int mmg_dropout_bwd(const float* dy, const void* keep, float* dx, long long n, float p, mmg_stream_t stream);

/* ---- depthwise 7x7 convolution, NHWC bf16 (ConvNeXt CNBlock) ------------------------------------------------ */

/* y = dwconv7x7(x; w, bias) (+ add), padding 3; x,y,add bf16 [n,H,W,C]; w fp32 tap-major [49][C]
 * (w[kh*7+kw][c] = weight[c,0,kh,kw]); flip != 0 reverses the taps (data gradient when x := dy).
 * C % 32 == 0.  Replaces CNBlock.block[0] of torchvision ConvNeXt (mmgclip/networks/encoder.py:53). */
int mmg_dwconv7_nhwc(const void* x, const float* w, const float* bias, const void* add, void* y, int n, int H, int W,
                     int C, int flip, mmg_stream_t stream);
/* The same contract on the matrix cores (ABI 5, csrc/dwconv7_mfma.hip): per channel and kernel row the 1-D convolution along x is a product with a
 * banded Toeplitz matrix (v_mfma_f32_16x16x32_bf16, taps rounded to bf16, fp32 accumulation); the NHWC <-> channel-plane layout changes run on the
 * LDS transposed reads.  With `add` the sum is rounded to bf16 twice (convolution, then + add).  (Rounds 1 - 2 had an earlier kernel of this name,
 * at parity with the VALU one and removed in ABI 4.) */
int mmg_dwconv7_nhwc_mfma(const void* x, const float* w, const float* bias, const void* add, void* y, int n, int H, int W,
                          int C, int flip, mmg_stream_t stream);
/* dw[49][C] += sum x(shifted) * dy ; dbias[C] += sum dy  (fp32, accumulated; dbias nullable) */
int mmg_dwconv7_wgrad(const void* x, const void* dy, float* dw, float* dbias, int n, int H, int W, int C,
                      mmg_stream_t stream);

/* ---- fused ConvNeXt block MLP (C in {96,128,192,256,384,512}) ---------------------------------------------------------- */

/* Number of bf16 elements of the packed weight image (8C^2 forward / backward=2, 12C^2 backward=1); 0 when C is unsupported. */
long long mmg_cnblock_packed_elems(int C, int backward);
/* Pack W1 fp32 [4C,C], W2 fp32 [C,4C] (torchvision CNBlock.block[3] / block[5]) into the per-chunk LDS images the fused
 * kernels stream (layout: csrc/cnblock_mlp.hip).  backward = 1 builds [W1 | gamma*W2^T | W1^T], backward = 2 builds [gamma*W2^T | W1^T]; both need gamma [C]. */
int mmg_cnblock_pack_weights(const float* w1, const float* w2, const float* gamma, void* packed, int C, int backward,
                             mmg_stream_t stream);
/* y = residual + gamma * ( GELU( LayerNorm(xd; ln_w, ln_b, eps) W1^T + b1 ) W2^T + b2 ), rows of [M,C] bf16.
 * Replaces CNBlock.block[2..5] + layer_scale + residual of torchvision ConvNeXt (mmgclip/networks/encoder.py:53) in one
 * launch: the 4C-wide hidden row stays in registers.  hpre (bf16 [M,4C], pre-GELU) and mean/rstd (fp32 [M]) are optional (all three or none)
 * outputs for a backward that does not recompute them; xln (bf16 [M,C], the LayerNorm output = operand of that backward's
 * weight-gradient GEMM; optional, only next to hpre) saves it a LayerNorm pass; gact (bf16 [M,4C], GELU(hidden) as the second GEMM
 * consumed it; optional, only next to hpre) is the operand of that backward's dW2 GEMM, which lets its data-gradient GEMM run
 * epilogue 5 (GELU' only) instead of 2.  hpre_kind = 1 (ABI 5; needs hpre and gact): `hpre` receives GELU'(hidden) instead of the hidden -
 * that data-gradient GEMM then runs epilogue 7, a multiply.  (ABI 2: xln added; ABI 4: gact added.) */
int mmg_cnblock_mlp_fwd(const void* xd, const float* ln_w, const float* ln_b, float eps, const void* packed,
                        const float* b1, const float* b2, const float* gamma, const void* residual, void* y, void* hpre,
                        void* xln, void* gact, float* mean, float* rstd, int hpre_kind, long long M, int C, mmg_stream_t stream);

/* Data path of the CNBlock MLP backward in one launch.  mmg_cnblock_mlp_bwd_supported(C): 1 (C in {96,128,192}) = the
 * hidden row h = LN(xd) W1^T + b1 is recomputed from the saved depthwise output, packed_bwd = pack(..., backward=1), hpre must
 * be NULL and the forward saves nothing 4C-wide; 2 (C = 384) = h is read back from the forward's hpre, packed_bwd =
 * pack(..., backward=2); 0 = unsupported.  Writes g = GELU(h), dh = (dy gamma W2) * GELU'(h) (bf16 [M,4C], operands of the
 * weight-gradient GEMMs dW2 = dy^T g, dW1 = dh^T xln), xln = LN(xd) and dxln = dh W1 (bf16 [M,C]) and the LN statistics for
 * mmg_layernorm_bwd.  With ln_dw / ln_db (fp32 [C], accumulated) the LayerNorm backward is fused into the epilogue: dxln then
 * receives d loss / d xd and no mmg_layernorm_bwd launch is needed.
 * (New capability: the reference never trains the image tower, mmgclip/networks/encoder.py:53.) */
int mmg_cnblock_mlp_bwd_supported(int C);
int mmg_cnblock_mlp_bwd(const void* dy, const void* xd, const float* ln_w, const float* ln_b, float eps,
                        const void* packed_bwd, const float* b1, const void* hpre, void* dh, void* g, void* xln,
                        void* dxln, float* mean, float* rstd, float* ln_dw, float* ln_db, long long M, int C,
                        mmg_stream_t stream);

/* CNBlock MLP backward with the weight gradients accumulated ON CHIP (round 3; C = 96, M a multiple of 64): nothing 4C-wide
 * reaches HBM.  Two launches (csrc/cnblock_bwdw.hip): (1) h = LN(xd) W1^T + b1, g = GELU(h), dW2raw += dy^T g, db2raw += colsum(dy);
 * (2) h again, dh = (dy gamma W2) * GELU'(h), dW1 += dh^T LN(xd), db1 += colsum(dh), d LN-out = dh W1, LayerNorm backward:
 * dd = d loss / d xd (bf16 [M,C]), ln_dw / ln_db += LayerNorm weight / bias gradients.  All fp32 outputs are ACCUMULATED (atomics);
 * dW2raw / db2raw are the un-scaled gradients mmg_layerscale_finalize expects (the same contract as dy^T g of the GEMM path).
 * `packed` / `b1f` come from mmg_cnblock_bwdw_pack (bf16 mmg_cnblock_bwdw_packed_elems(C) elements; fp32 [4C]): the W1 LDS image,
 * the LayerNorm-folded W1 and layer-scale-folded W2^T as per-wave MFMA fragments, and b1' = b1 + W1 ln_b.
 * Replaces autograd of CNBlock.block[2..5] + layer_scale (torchvision ConvNeXt; the reference runs it frozen, encoder.py:53). */
int mmg_cnblock_bwdw_supported(int C);
long long mmg_cnblock_bwdw_packed_elems(int C);
int mmg_cnblock_bwdw_pack(const float* w1, const float* w2, const float* ln_w, const float* ln_b, const float* layer_scale,
                          const float* b1, void* packed, float* b1f, int C, mmg_stream_t stream);
int mmg_cnblock_bwdw(const void* dy, const void* xd, const float* ln_w, const float* ln_b, float eps, const void* packed,
                     const float* b1f, void* dd, float* dW1, float* db1, float* dW2raw, float* db2raw, float* ln_dw,
                     float* ln_db, long long M, int C, mmg_stream_t stream);

/* ---- AveragedMedicalCLIPLoss helpers (reference mmgclip/loss/losses.py:98-216) ------------------------------------------------ */

/* `_assign_labels` (losses.py:148-162) on the device: walking i = 0..n-1, an unlabelled text opens the next cluster and every
 * still unlabelled j > i with sim[i][j] >= threshold joins it.  sim fp32 [n,n] (ld); labels int64 [n]; counts int32 [n]
 * (members per cluster, first *k_out entries valid); k_out int32 device scalar = number of clusters.  n <= 16384. */
int mmg_greedy_threshold_labels(const float* sim, int ld, int n, float threshold, long long* labels, int* counts, int* k_out,
                                mmg_stream_t stream);

/* `_average_logits` (losses.py:164-186): out[i,c] = mean_{j: labels[j]==c} logits[i,j]  (logits fp32 [n,N], out fp32 [n,k]),
 * and its backward dlogits[i,j] = dout[i,labels[j]] / counts[labels[j]]. */
int mmg_cluster_mean_cols_fwd(const float* logits, int ld, int n, int N, const long long* labels, const int* counts, int k,
                              float* out, int ldo, mmg_stream_t stream);
int mmg_cluster_mean_cols_bwd(const float* dout, int ldo, int n, int N, const long long* labels, const int* counts,
                              float* dlogits, int ld, mmg_stream_t stream);

/* ---- fp8 (OCP e4m3) forward GEMM path: BASELINE config C5 "ConvNeXt-base fp8 MFMA path" ---------------------------------
 * The reference has no reduced-precision path (its towers run torch fp32, mmgclip/networks/encoder.py:53,156); these replace
 * the same nn.Linear forwards of torchvision's CNBlock as mmg_gemm_nt_bf16 does, on v_mfma_f32_16x16x128_f8f6f4 (twice the
 * bf16 MFMA rate, half the operand bytes).  Backward stays bf16 (saved pre-activations and bf16 weight copies). */

/* C[M,N] = epilogue( alpha * (alpha_dev ? *alpha_dev : 1) * A[M,K] B[N,K]^T + bias ): A, B e4m3 bytes row-major (K contiguous,
 * lda / ldb in bytes, multiples of 16), fp32 accumulate.  Epilogue as mmg_gemm_nt_bf16 with epi in {0 none, 1 GELU, 3 ReLU, 6 GELU + GELU' side output};
 * out_kind: 0 bf16 | 1 fp32 | 2 e4m3 bytes (saturating at +-448; ldc in elements of that type).  K % 128 == 0, N % 8 == 0.
 * alpha_dev: device scalar, e.g. scales[1] of mmg_quantize_e4m3_f32 (so weight scales never visit the host). */
int mmg_gemm_nt_fp8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                    const float* bias, const float* colscale, const void* residual, int ldr, void* aux_out, int ldao,
                    int epi, int out_kind, float alpha, const float* alpha_dev, mmg_stream_t stream);

/* ---- fp8 backward of the ConvNeXt blocks (ABI 5; BASELINE config C5 "ConvNeXt-base fp8 MFMA path"; the reference itself has no fp8) ---------- */

/* src bf16 [n] -> dst OCP e5m2 bytes [n] with the per-tensor power-of-two scale 2^floor(log2(16384 / max|src|)) computed on the device
 * (amax fp32 [1] is scratch); scales fp32 [2] = (scale, 1 / scale).  n % 8 == 0.  The gradient entering a CNBlock backward
 * (torchvision CNBlock behind mmgclip/networks/encoder.py:53). */
int mmg_quantize_e5m2_bf16(const void* src, long long n, float* amax, void* dst, float* scales, mmg_stream_t stream);
/* The same in ONE pass ("delayed scaling"): scale = 2^floor(log2(4096 / *amax_prev)) from the absmax this tensor had at its previous quantisation
 * (1 when that is 0), and the tensor's own absmax is left in amax_next (fp32 [1]) for the next call. */
int mmg_quantize_e5m2_bf16_delayed(const void* src, long long n, const float* amax_prev, float* amax_next, void* dst, float* scales,
                                   mmg_stream_t stream);
/* The delayed cast of a gradient MATRIX src bf16 [M, C] (contiguous) and colsum[c] += sum_m src[m][c] (fp32 [C]) in one pass over src: the bias gradient
 * of a CNBlock's second Linear is taken from the bf16 gradient, not from its cast.  C = 16 x a divisor of 256 (256, 512, 1024 ... of ConvNeXt-B). */
int mmg_quantize_e5m2_colsum_bf16(const void* src, int M, int C, const float* amax_prev, float* amax_next, void* dst, float* scales,
                                  float* colsum, mmg_stream_t stream);
/* C[M,N] = epilogue( alpha * alpha_dev * alpha_dev2 * A[M,K] B[N,K]^T ): A = e5m2 (a_e5m2 != 0) or e4m3 bytes, B e4m3 bytes, fp32 accumulate on the
 * K = 128 MFMA; epi 0 none | 5 multiply by GELU'(aux_in) | 7 multiply by aux_in (aux_in bf16 [M,N]); C bf16 / fp32 / e5m2 bytes (out_kind 0 / 1 / 3).
 * The two data-gradient GEMMs of a CNBlock (dh = (dy (gamma W2)) * GELU'(h) handed on in 8 bits, d LN-out = dh W1).  K % 128 == 0. */
int mmg_gemm_nt_fp8_bwd(const void* A, int lda, int a_e5m2, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                        const void* aux_in, int ldai, int epi, int out_kind, float alpha, const float* alpha_dev,
                        const float* alpha_dev2, mmg_stream_t stream);
/* C[N1,N2] (fp32) += alpha * alpha_dev * A[M,N1]^T B[M,N2]: A e5m2 (a_e5m2 != 0) or e4m3 bytes, B e4m3 bytes, both row-major with the reduction index
 * slow (transposed 8-bit fragment reads, ds_read_b64_tr_b8); colsum_a [N1] (optional) += alpha * alpha_dev * column sums of A.  N1, N2, lda, ldb
 * multiples of 16.  The two weight-gradient GEMMs of a CNBlock on the 8-bit operands its forward / data-gradient GEMMs already hold. */
int mmg_gemm_tn_fp8(const void* A, int lda, int a_e5m2, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                    float alpha, const float* alpha_dev, float* colsum_a, mmg_stream_t stream);

/* amax[0] = max(amax[0], max |src[i]|) (caller zeroes amax);  then  dst = e4m3(src * scale) with the power-of-two
 * scale = 2^floor(log2(448 / amax)) (1 when amax is null / 0), scales[0] = scale, scales[1] = 1 / scale.  n % 4 == 0. */
int mmg_absmax_f32(const float* src, long long n, float* amax, mmg_stream_t stream);
int mmg_quantize_e4m3_f32(const float* src, long long n, const float* amax, void* dst, float* scales, mmg_stream_t stream);

/* mmg_layernorm_fwd whose output row is written as e4m3 bytes (unscaled, saturating; ldy in bytes, multiple of 8). */
int mmg_layernorm_fwd_fp8(const void* x, int ldx, const float* gamma, const float* beta, float eps, void* y8, int ldy,
                          float* mean, float* rstd, int M, int C, mmg_stream_t stream);

/* ---- ResNet-50 tower pieces (reference ResNet50Encoder, mmgclip/networks/encoder.py:57-119: torchvision resnet50 minus fc,
 * everything frozen except layer4).  NHWC bf16; convolutions = im2col + mmg_gemm_nt_bf16 (1x1: no im2col at all). ------------ */

/* col[n*Ho*Wo, Kp] (bf16) from x[n,H,W,C]: column (kh*KW + kw)*C + c, zeros outside the image and in the K padding; C % 8 == 0. */
int mmg_im2col_nhwc(const void* x, void* col, int n, int H, int W, int C, int KH, int KW, int stride, int pad, int Kp,
                    mmg_stream_t stream);
/* dx[n,H,W,C] = adjoint of mmg_im2col_nhwc applied to dcol (data gradient of a k x k convolution); gather, no atomics. */
int mmg_col2im_nhwc(const void* dcol, void* dx, int n, int H, int W, int C, int KH, int KW, int stride, int pad, int Kp,
                    mmg_stream_t stream);
/* nn.MaxPool2d(3, stride 2, padding 1) forward (encoder.py:107). */
int mmg_maxpool3x3s2_nhwc(const void* x, void* y, int n, int H, int W, int C, mmg_stream_t stream);
/* nn.BatchNorm2d on rows [M = n*H*W, C]: column sums of x and x^2 (fp32, accumulated; zero them first) ... */
int mmg_bn_stats(const void* x, int M, int C, float* sum, float* sumsq, mmg_stream_t stream);
/* ... -> mean / rstd and the fused affine (scale, shift).  train != 0: batch statistics (biased variance) and torch's
 * running-statistics update with `momentum` (unbiased variance); train == 0: the running statistics are used. */
int mmg_bn_finalize(const float* sum, const float* sumsq, int M, int C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, int train, float* mean, float* rstd,
                    float* scale, float* shift, mmg_stream_t stream);
/* y = x * scale[c] + shift[c] (+ residual) (ReLU when relu != 0): bn + the bottleneck's shortcut add + ReLU in one pass. */
int mmg_bn_apply(const void* x, const float* scale, const float* shift, const void* residual, void* y, int M, int C, int relu,
                 mmg_stream_t stream);
/* Training-mode backward of y = relu?(bn(x) (+ residual)): g = dy masked by `out` > 0 (the layer's own output; NULL = no ReLU).
 * reduce: sum_g[c] += g, sum_gx[c] += g * xhat (= d beta, d gamma).  apply: dx = gamma rstd (g - sum_g/M - xhat sum_gx/M);
 * dres (nullable) = g, the gradient of the residual branch; dgamma / dbeta (nullable pair, fp32 [C]) += sum_gx / sum_g. */
int mmg_bn_bwd_reduce(const void* dy, const void* x, const void* out, const float* mean, const float* rstd, int M, int C,
                      float* sum_g, float* sum_gx, mmg_stream_t stream);
int mmg_bn_bwd_apply(const void* dy, const void* x, const void* out, const float* mean, const float* rstd, const float* gamma,
                     const float* sum_g, const float* sum_gx, int M, int C, void* dx, void* dres, float* dgamma, float* dbeta,
                     mmg_stream_t stream);

/* ---- BERT attention / embeddings / pooling -------------------------------------------------------------------- */

/* ctx[B*S,Hd] = per-head softmax(Q K^T * scale + key mask) V with qkv = [B*S, q|k|v] bf16 (head h at columns h*64
 * of each third); mask int64 [B,S] (1 attend / 0 pad, nullable); lse fp32 [B,heads,S] (nullable) for the backward.
 * head_dim 64, S <= 512.  Replaces HF BertSelfAttention (mmgclip/networks/encoder.py:156). */
int mmg_attention_fwd(const void* qkv, int ld, const long long* mask, void* ctx, int ldc, float* lse, int B, int S,
                      int heads, int Hd, float scale, mmg_stream_t stream);
/* dqkv (same layout as qkv) from dctx; recomputes probabilities from lse.  S <= 256. */
int mmg_attention_bwd(const void* qkv, int ld, const long long* mask, const void* ctx, int ldc, const float* lse,
                      const void* dctx, int lddc, void* dqkv, int lddq, int B, int S, int heads, int Hd, float scale,
                      mmg_stream_t stream);

/* Flash-style tiled attention for sequences of any length (ViT-B/16 on 1024x1024: S = 4097; BERT training at S > 256):
 * same layout and semantics as mmg_attention_fwd/bwd, K/V (or Q/dO) streamed through LDS in 64-row tiles, online
 * softmax, no S x S tensor.  delta_ws: caller-provided fp32 [B*heads*S] scratch. */
int mmg_attention_long_fwd(const void* qkv, int ld, const long long* mask, void* ctx, int ldc, float* lse, int B, int S,
                           int heads, int Hd, float scale, mmg_stream_t stream);
int mmg_attention_long_bwd(const void* qkv, int ld, const long long* mask, const void* ctx, int ldc, const float* lse,
                           const void* dctx, int lddc, void* dqkv, int lddq, float* delta_ws, int B, int S, int heads,
                           int Hd, float scale, mmg_stream_t stream);

/* The same attention on the packed ("unpadded") layout: sequence b occupies rows cu_seqlens[b] .. cu_seqlens[b+1] of qkv / ctx /
 * dctx / dqkv (int32 [B+1], device), every token attended, lengths in [1, S_max]; lse is [B, heads, S_max].  The reference pads
 * every prompt to max_length (mmgclip/dataset/dataset.py:347) and reads only the [SEP] row (mmgclip_model.py:110-111): the
 * text tower runs on the valid tokens only and scatters its last hidden state back into the padded layout. */
int mmg_attention_varlen_fwd(const void* qkv, int ld, const int* cu_seqlens, void* ctx, int ldc, float* lse, int B, int S_max,
                             int heads, int Hd, float scale, mmg_stream_t stream);
int mmg_attention_varlen_bwd(const void* qkv, int ld, const int* cu_seqlens, const void* ctx, int ldc, const float* lse,
                             const void* dctx, int lddc, void* dqkv, int lddq, int B, int S_max, int heads, int Hd,
                             float scale, mmg_stream_t stream);

/* Training-mode dropout of the text tower.  The reference runs HF BertModel under model.train()
 * (mmgclip/experiments/ClassifierExperiment.py:97 -> mmgclip/networks/encoder.py:156) with hidden_dropout_prob =
 * attention_probs_dropout_prob = 0.1 (notebooks/bert_experimental.ipynb:609-624).  Masks are counter-based (csrc/dropout.h;
 * restated in oracle/dropout_oracle.py): element `index` of dropout site `site` is kept, and scaled by 1/(1-p), iff
 * fmix32(index * 0x9E3779B1 + key(seed, site)) >= floor(p * 2^32) - the backward regenerates the mask from (p, seed, site).
 *   mmg_attention_dropout_fwd/_bwd: the whole-sequence attention above (S <= 512 forward, <= 256 backward; _long_bwd beyond) with the
 *     probabilities dropped before P V; cu_seqlens != NULL selects the packed layout (then `mask` is unused), else the padded one.
 *     index = (((first_sequence + b) * heads + h) * 512 + query) * 512 + key  (first_sequence: position of this call's
 *     sequence 0 in the whole batch, so that micro-batches of one batch draw disjoint masks).
 *   mmg_dropout_f32 : x fp32 [M,C] <- dropout(x) in place, optional bf16 copy xb.    index = token * C + column, token = rows[m]
 *   mmg_dropout_bf16: out bf16 [M,C] = dropout(in)  (the same mask on a gradient).    (rows NULL: token = m).
 * p in [0,1); p = 0 is the identity. */
int mmg_attention_dropout_fwd(const void* qkv, int ld, const long long* mask, const int* cu_seqlens, void* ctx, int ldc,
                              float* lse, int B, int S, int heads, int Hd, float scale, float p, unsigned long long seed,
                              unsigned site, int first_sequence, mmg_stream_t stream);
int mmg_attention_dropout_bwd(const void* qkv, int ld, const long long* mask, const int* cu_seqlens, const void* ctx, int ldc,
                              const float* lse, const void* dctx, int lddc, void* dqkv, int lddq, int B, int S, int heads,
                              int Hd, float scale, float p, unsigned long long seed, unsigned site, int first_sequence,
                              mmg_stream_t stream);
/* the same backward for 256 < S <= 512 (padded layout only): tiled dQ / dK,dV kernels; delta_ws = fp32 [B * heads * S] workspace */
int mmg_attention_dropout_long_bwd(const void* qkv, int ld, const long long* mask, const void* ctx, int ldc, const float* lse,
                                   const void* dctx, int lddc, void* dqkv, int lddq, float* delta_ws, int B, int S, int heads,
                                   int Hd, float scale, float p, unsigned long long seed, unsigned site, int first_sequence,
                                   mmg_stream_t stream);
int mmg_dropout_f32(float* x, int ldx, void* xb, int ldb, const long long* rows, long long M, int C, float p,
                    unsigned long long seed, unsigned site, mmg_stream_t stream);
int mmg_dropout_bf16(const void* in, int ldi, void* out, int ldo, const long long* rows, long long M, int C, float p,
                     unsigned long long seed, unsigned site, mmg_stream_t stream);

/* out[m,:] = word[ids[m]] + pos[m % S] + type[type_ids[m]] (bf16 tables [V|P|T, H]); HF BertEmbeddings before its
 * LayerNorm (mmgclip/networks/encoder.py:156). */
int mmg_bert_embed_fwd(const long long* ids, const long long* type_ids, const void* word, const void* pos,
                       const void* type, void* out, int M, int S, int H, int V, int T, mmg_stream_t stream);
/* table gradients (fp32, accumulated) from g = d out [B*S,H] bf16 */
int mmg_bert_embed_bwd(const void* g, const long long* ids, const long long* type_ids, float* dword, float* dpos,
                       float* dtype, int B, int S, int H, int V, int T, mmg_stream_t stream);

/* EOS pooling: out[b,:] (fp32) = hidden[b, sum(mask[b])-1, :]; idx_out (int32 [B], nullable) keeps the index.
 * Replaces mmgclip/networks/mmgclip_model.py:110-111. */
int mmg_eos_pool_fwd(const void* hidden, const long long* mask, float* out, int* idx_out, int B, int S, int H,
                     mmg_stream_t stream);
/* the same from an fp32 hidden state [B*S, H] (the text tower's fp32 residual stream) */
int mmg_eos_pool_fwd_f32(const float* hidden, const long long* mask, float* out, int* idx_out, int B, int S, int H,
                         mmg_stream_t stream);
int mmg_eos_pool_bwd(const float* dout, const int* idx, void* dhidden, int B, int S, int H, mmg_stream_t stream);

/* ViT token assembly (torchvision VisionTransformer: class token + patch tokens + encoder.pos_embedding):
 * out[b,0,:] = cls + pos[0]; out[b,1+p,:] = tok[b*Np+p,:] + pos[1+p]; S = Np + 1; all bf16. */
int mmg_vit_assemble_fwd(const void* tok, const void* cls, const void* pos, void* out, int B, int S, int H,
                         mmg_stream_t stream);
/* dtok (bf16) = g rows 1..; dpos[S,H] += sum_b g; dcls[H] += sum_b g[b,0,:]  (fp32, accumulated) */
int mmg_vit_assemble_bwd(const void* g, void* dtok, float* dpos, float* dcls, int B, int S, int H, mmg_stream_t stream);
/* out[b,:] (fp32) = hidden[b*S + idx[b], :] — pooling at explicit token indices (class token: idx = 0);
 * its backward is mmg_eos_pool_bwd. */
int mmg_gather_rows_fwd(const void* hidden, const int* idx, float* out, int B, int S, int H, mmg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMGCLIP_HIP_H */
