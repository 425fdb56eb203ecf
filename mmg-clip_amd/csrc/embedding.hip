// BERT embeddings (gfx950): gather word + position + token-type rows and their gradients.
// Replaces HF BertEmbeddings (3 nn.Embedding lookups + add, before its LayerNorm) reached from
// mmgclip/networks/encoder.py:156; table sizes from notebooks/bert_experimental.ipynb:609-624.
// HBM-bound row gathers: one wave per token row, 16 bytes per lane.
#include "common.h"

__device__ __forceinline__ void unpack8e(const uint4 v, float* f) {
    f[0] = bf2f_lo(v.x); f[1] = bf2f_hi(v.x); f[2] = bf2f_lo(v.y); f[3] = bf2f_hi(v.y);
    f[4] = bf2f_lo(v.z); f[5] = bf2f_hi(v.z); f[6] = bf2f_lo(v.w); f[7] = bf2f_hi(v.w);
}

// out[m,:] = word[ids[m],:] + pos[m % S,:] + type[tt[m],:]     (bf16 tables, fp32 add, bf16 out)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ ids, const long long* __restrict__ tt,
                                                        const bf16_t* __restrict__ word, const bf16_t* __restrict__ pos,
                                                        const bf16_t* __restrict__ type, bf16_t* __restrict__ out, int M,
                                                        int S, int H, int V, int T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = H / 8;
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        long long id = ids[m];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        long long t = tt ? tt[m] : 0;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const int p = m % S;
        for (int c = lane; c < nch; c += 64) {
            float a[8], b[8], d[8];
            unpack8e(*reinterpret_cast<const uint4*>(word + (size_t)id * H + c * 8), a);
            unpack8e(*reinterpret_cast<const uint4*>(pos + (size_t)p * H + c * 8), b);
            unpack8e(*reinterpret_cast<const uint4*>(type + (size_t)t * H + c * 8), d);
            uint4 o;
            o.x = pack2bf(a[0] + b[0] + d[0], a[1] + b[1] + d[1]);
            o.y = pack2bf(a[2] + b[2] + d[2], a[3] + b[3] + d[3]);
            o.z = pack2bf(a[4] + b[4] + d[4], a[5] + b[5] + d[5]);
            o.w = pack2bf(a[6] + b[6] + d[6], a[7] + b[7] + d[7]);
            *reinterpret_cast<uint4*>(out + (size_t)m * H + c * 8) = o;
        }
    }
}

MMG_API int mmg_bert_embed_fwd(const long long* ids, const long long* type_ids, const void* word, const void* pos,
                               const void* type, void* out, int M, int S, int H, int V, int T, hipStream_t stream) {
    MMG_CHECK_ARG(ids && word && pos && type && out && M > 0 && S > 0 && H % 8 == 0 && V > 0 && T > 0,
                  "mmg_bert_embed_fwd: bad argument");
    int blocks = cdiv(M, 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(blocks), dim3(256), 0, stream, ids, type_ids, (const bf16_t*)word,
                       (const bf16_t*)pos, (const bf16_t*)type, (bf16_t*)out, M, S, H, V, T);
    MMG_LAUNCH_CHECK("mmg_bert_embed_fwd");
    return 0;
}

// dword[ids[m],:] += g[m,:]  (fp32 atomics; one wave per token = 256-byte contiguous atomic rows)
__global__ __launch_bounds__(256) void embed_bwd_word_kernel(const bf16_t* __restrict__ g, const long long* __restrict__ ids,
                                                             float* __restrict__ dword, int M, int H, int V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        long long id = ids[m];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        for (int c = lane; c < H; c += 64) atomicAdd(dword + (size_t)id * H + c, bf2f(g[(size_t)m * H + c]));
    }
}

// dpos[p,:] += sum_b g[b*S+p,:]   (one block per position and 256-column slab: no atomics between sequences)
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const bf16_t* __restrict__ g, float* __restrict__ dpos, int B, int S,
                                                            int H) {
    const int p = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= H) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += bf2f(g[((size_t)b * S + p) * H + c]);
    dpos[(size_t)p * H + c] += s;
}

// dtype[t,:] += sum_{m: tt[m]==t} g[m,:]  (block-local reduction in registers over a token chunk, then atomics)
__global__ __launch_bounds__(256) void embed_bwd_type_kernel(const bf16_t* __restrict__ g, const long long* __restrict__ tt,
                                                             float* __restrict__ dtype, int M, int H, int T,
                                                             int rows_per_block) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= H) return;
    const int m0 = blockIdx.x * rows_per_block, m1 = min(m0 + rows_per_block, M);
    float s0 = 0.f, s1 = 0.f;
    for (int m = m0; m < m1; ++m) {
        const float v = bf2f(g[(size_t)m * H + c]);
        const long long t = tt ? tt[m] : 0;
        if (t <= 0) s0 += v; else s1 += v;
    }
    atomicAdd(dtype + c, s0);
    if (T > 1) atomicAdd(dtype + H + c, s1);
}

// Gradients of the three tables given g = d(sum of embeddings) [M,H] bf16.  All outputs accumulate (fp32).
MMG_API int mmg_bert_embed_bwd(const void* g, const long long* ids, const long long* type_ids, float* dword, float* dpos,
                               float* dtype, int B, int S, int H, int V, int T, hipStream_t stream) {
    MMG_CHECK_ARG(g && ids && dword && dpos && dtype && B > 0 && S > 0 && H > 0 && V > 0 && T > 0 && T <= 2,
                  "mmg_bert_embed_bwd: bad argument (token types > 2 unsupported)");
    const int M = B * S;
    int blocks = cdiv(M, 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_bwd_word_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)g, ids, dword, M, H, V);
    hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3(S, cdiv(H, 256)), dim3(256), 0, stream, (const bf16_t*)g, dpos, B, S, H);
    const int rpb = 128;
    hipLaunchKernelGGL(embed_bwd_type_kernel, dim3(cdiv(M, rpb), cdiv(H, 256)), dim3(256), 0, stream, (const bf16_t*)g,
                       type_ids, dtype, M, H, T, rpb);
    MMG_LAUNCH_CHECK("mmg_bert_embed_bwd");
    return 0;
}

// out[b,:] = hidden[b*S + idx_b, :] with idx_b = sum(mask[b,:]) - 1   (EOS pooling, mmgclip_model.py:110-111); fp32 out
template <bool F32>
__global__ __launch_bounds__(256) void eos_pool_kernel(const void* __restrict__ hidden_, const long long* __restrict__ mask,
                                                       float* __restrict__ out, int* __restrict__ idx_out, int S, int H) {
    const bf16_t* hidden = reinterpret_cast<const bf16_t*>(hidden_);
    const float* hidden32 = reinterpret_cast<const float*>(hidden_);
    __shared__ int s_idx;
    const int b = blockIdx.x;
    if (threadIdx.x < 64) {
        int cnt = 0;
        for (int k = threadIdx.x; k < S; k += 64) cnt += mask[(size_t)b * S + k] != 0 ? (int)mask[(size_t)b * S + k] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (threadIdx.x == 0) {
            int i = cnt - 1;
            if (i < 0) i += S;            // torch negative indexing: -1 -> last position
            s_idx = i;
            if (idx_out) idx_out[b] = i;
        }
    }
    __syncthreads();
    const int i = s_idx;
    for (int c = threadIdx.x; c < H; c += 256)
        out[(size_t)b * H + c] = F32 ? hidden32[((size_t)b * S + i) * H + c] : bf2f(hidden[((size_t)b * S + i) * H + c]);
}
MMG_API int mmg_eos_pool_fwd(const void* hidden, const long long* mask, float* out, int* idx_out, int B, int S, int H,
                             hipStream_t stream) {
    MMG_CHECK_ARG(hidden && mask && out && B > 0 && S > 0 && H > 0, "mmg_eos_pool_fwd: bad argument");
    hipLaunchKernelGGL(eos_pool_kernel<false>, dim3(B), dim3(256), 0, stream, hidden, mask, out, idx_out, S, H);
    MMG_LAUNCH_CHECK("mmg_eos_pool_fwd");
    return 0;
}
// the same gather from an fp32 hidden state (the text tower hands over its fp32 residual stream: no rounding before the projection)
MMG_API int mmg_eos_pool_fwd_f32(const float* hidden, const long long* mask, float* out, int* idx_out, int B, int S, int H,
                                 hipStream_t stream) {
    MMG_CHECK_ARG(hidden && mask && out && B > 0 && S > 0 && H > 0, "mmg_eos_pool_fwd_f32: bad argument");
    hipLaunchKernelGGL(eos_pool_kernel<true>, dim3(B), dim3(256), 0, stream, (const void*)hidden, mask, out, idx_out, S, H);
    MMG_LAUNCH_CHECK("mmg_eos_pool_fwd_f32");
    return 0;
}
// dhidden = 0 except row idx_b of each sequence = dout[b,:]  (bf16)
__global__ __launch_bounds__(256) void eos_pool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ idx,
                                                           bf16_t* __restrict__ dh, int S, int H) {
    const int row = blockIdx.x;          // b*S + s
    const int b = row / S, s = row - b * S;
    const bool hit = (idx[b] == s);
    for (int c = threadIdx.x; c < H; c += 256) dh[(size_t)row * H + c] = hit ? f2bf(dout[(size_t)b * H + c]) : (bf16_t)0;
}
MMG_API int mmg_eos_pool_bwd(const float* dout, const int* idx, void* dhidden, int B, int S, int H, hipStream_t stream) {
    MMG_CHECK_ARG(dout && idx && dhidden && B > 0 && S > 0 && H > 0, "mmg_eos_pool_bwd: bad argument");
    hipLaunchKernelGGL(eos_pool_bwd_kernel, dim3(B * S), dim3(256), 0, stream, dout, idx, (bf16_t*)dhidden, S, H);
    MMG_LAUNCH_CHECK("mmg_eos_pool_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// ViT token assembly: X[b, 0, :] = cls + pos[0], X[b, 1+p, :] = tok[b*Np + p, :] + pos[1+p]   (bf16 in/out, fp32 add)
// torchvision VisionTransformer._process_input + class_token concat + encoder.pos_embedding add.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_assemble_kernel(const bf16_t* __restrict__ tok, const bf16_t* __restrict__ cls,
                                                           const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, int B,
                                                           int S, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = H / 8, Np = S - 1;
    for (int m = blockIdx.x * 4 + wave; m < B * S; m += gridDim.x * 4) {
        const int b = m / S, s = m - b * S;
        const bf16_t* src = s == 0 ? cls : tok + ((size_t)b * Np + (s - 1)) * H;
        for (int c = lane; c < nch; c += 64) {
            float a[8], p[8];
            unpack8e(*reinterpret_cast<const uint4*>(src + c * 8), a);
            unpack8e(*reinterpret_cast<const uint4*>(pos + (size_t)s * H + c * 8), p);
            uint4 o;
            o.x = pack2bf(a[0] + p[0], a[1] + p[1]); o.y = pack2bf(a[2] + p[2], a[3] + p[3]);
            o.z = pack2bf(a[4] + p[4], a[5] + p[5]); o.w = pack2bf(a[6] + p[6], a[7] + p[7]);
            *reinterpret_cast<uint4*>(out + (size_t)m * H + c * 8) = o;
        }
    }
}
MMG_API int mmg_vit_assemble_fwd(const void* tok, const void* cls, const void* pos, void* out, int B, int S, int H,
                                 hipStream_t stream) {
    MMG_CHECK_ARG(tok && cls && pos && out && B > 0 && S > 1 && H % 8 == 0, "mmg_vit_assemble_fwd: bad argument");
    int blocks = cdiv(B * S, 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)tok, (const bf16_t*)cls,
                       (const bf16_t*)pos, (bf16_t*)out, B, S, H);
    MMG_LAUNCH_CHECK("mmg_vit_assemble_fwd");
    return 0;
}
// backward: dtok[b*Np+p,:] = g[b,1+p,:] (bf16 copy) ; dpos[s,:] += sum_b g[b,s,:] ; dcls[:] += sum_b g[b,0,:]
__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const bf16_t* __restrict__ g, bf16_t* __restrict__ dtok,
                                                               float* __restrict__ dpos, float* __restrict__ dcls, int B, int S,
                                                               int H) {
    const int s = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= H) return;
    const int Np = S - 1;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const bf16_t v = g[((size_t)b * S + s) * H + c];
        acc += bf2f(v);
        if (s > 0) dtok[((size_t)b * Np + (s - 1)) * H + c] = v;
    }
    dpos[(size_t)s * H + c] += acc;
    if (s == 0) dcls[c] += acc;
}
MMG_API int mmg_vit_assemble_bwd(const void* g, void* dtok, float* dpos, float* dcls, int B, int S, int H,
                                 hipStream_t stream) {
    MMG_CHECK_ARG(g && dtok && dpos && dcls && B > 0 && S > 1 && H > 0, "mmg_vit_assemble_bwd: bad argument");
    hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3(S, cdiv(H, 256)), dim3(256), 0, stream, (const bf16_t*)g, (bf16_t*)dtok,
                       dpos, dcls, B, S, H);
    MMG_LAUNCH_CHECK("mmg_vit_assemble_bwd");
    return 0;
}
// out[b,:] (fp32) = hidden[b*S + idx[b], :]  (row gather with explicit indices; ViT class-token pooling uses idx = 0)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ hidden, const int* __restrict__ idx,
                                                          float* __restrict__ out, int S, int H) {
    const int b = blockIdx.x;
    const int i = idx[b];
    for (int c = threadIdx.x; c < H; c += 256) out[(size_t)b * H + c] = bf2f(hidden[((size_t)b * S + i) * H + c]);
}
MMG_API int mmg_gather_rows_fwd(const void* hidden, const int* idx, float* out, int B, int S, int H, hipStream_t stream) {
    MMG_CHECK_ARG(hidden && idx && out && B > 0 && S > 0 && H > 0, "mmg_gather_rows_fwd: bad argument");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, stream, (const bf16_t*)hidden, idx, out, S, H);
    MMG_LAUNCH_CHECK("mmg_gather_rows_fwd");
    return 0;
}
