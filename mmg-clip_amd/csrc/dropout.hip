// Element-wise dropout of the text tower's hidden states (see dropout.h for the mask and why it is counter-based).
// HF BertModel applies nn.Dropout(hidden_dropout_prob) after the embedding LayerNorm, on the attention-output projection and
// on the FFN output, each before the residual add (transformers BertEmbeddings / BertSelfOutput / BertOutput; the reference
// reaches them through mmgclip/networks/encoder.py:156 in training mode).  These tensors are [tokens, 768]: HBM-bound, 4 + 4
// (fp32 in place) or 2 + 2 (bf16) bytes per element, 16-byte accesses; the three 32-bit multiplies of the hash are noise.
#include "common.h"
#include "dropout.h"

struct DropRows {
    const long long* rows;      // nullable: token id (b * S + s) of every row (packed layout)
    long M; int C;
    DropArgs d;
};

__device__ __forceinline__ float drop1(float v, unsigned index, const DropArgs d) {
    return mmg_drop_bits(index, d.key) >= d.thresh ? v * d.scale : 0.f;
}

// x fp32 [M,C] in place; optional bf16 copy of the result
__global__ __launch_bounds__(256) void dropout_f32_kernel(float* __restrict__ x, int ldx, bf16_t* __restrict__ xb, int ldb,
                                                          const DropRows a) {
    const int c4 = a.C / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.M * c4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        const unsigned tok = (unsigned)(a.rows ? a.rows[r] : r);
        const unsigned base = tok * (unsigned)a.C + (unsigned)c;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = drop1(v[e], base + e, a.d);
        *reinterpret_cast<f32x4*>(x + r * ldx + c) = v;
        if (xb) {
            uint2 o;
            o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(xb + r * ldb + c) = o;
        }
    }
}

// out bf16 [M,C] = dropout(in bf16 [M,C])   (gradient path: the same mask applied to the incoming gradient)
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16_t* __restrict__ in, int ldi, bf16_t* __restrict__ out, int ldo,
                                                           const DropRows a) {
    const int c8 = a.C / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.M * c8; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        const unsigned tok = (unsigned)(a.rows ? a.rows[r] : r);
        const unsigned base = tok * (unsigned)a.C + (unsigned)c;
        const uint4 raw = *reinterpret_cast<const uint4*>(in + r * ldi + c);
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = pack2bf(drop1(bf2f_lo(w[e]), base + 2 * e, a.d), drop1(bf2f_hi(w[e]), base + 2 * e + 1, a.d));
        *reinterpret_cast<uint4*>(out + r * ldo + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

static int drop_args(const char* who, float p, unsigned long long seed, unsigned site, DropArgs& d) {
    MMG_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability %g must be in [0, 1)", who, (double)p);
    d.key = mmg_drop_key(seed, site);
    d.thresh = mmg_drop_threshold(p);
    d.scale = 1.0f / (1.0f - p);
    return 0;
}

// x <- dropout(x) in place (fp32 [M,C], row stride ldx); xb (nullable) receives the bf16 copy.  rows (nullable, int64 [M]) maps
// row -> token id for the mask index.  Element (row, c) is kept iff mmg_drop_bits(token * C + c, key(seed, site)) >= p * 2^32.
MMG_API int mmg_dropout_f32(float* x, int ldx, void* xb, int ldb, const long long* rows, long long M, int C, float p,
                            unsigned long long seed, unsigned site, hipStream_t stream) {
    MMG_CHECK_ARG(x && M > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0, "mmg_dropout_f32: bad x / shape (M=%lld C=%d ldx=%d)", M, C, ldx);
    MMG_CHECK_ARG(!xb || (ldb >= C && ldb % 4 == 0), "mmg_dropout_f32: bad bf16 leading dimension %d", ldb);
    DropRows a = {rows, (long)M, C, {}};
    if (drop_args("mmg_dropout_f32", p, seed, site, a.d)) return 1;
    const long work = (long)M * (C / 4);
    hipLaunchKernelGGL(dropout_f32_kernel, dim3((unsigned)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192)), dim3(256), 0, stream, x, ldx,
                       (bf16_t*)xb, ldb, a);
    MMG_LAUNCH_CHECK("mmg_dropout_f32");
    return 0;
}

MMG_API int mmg_dropout_bf16(const void* in, int ldi, void* out, int ldo, const long long* rows, long long M, int C, float p,
                             unsigned long long seed, unsigned site, hipStream_t stream) {
    MMG_CHECK_ARG(in && out && M > 0 && C > 0 && C % 8 == 0 && ldi >= C && ldo >= C && ldi % 8 == 0 && ldo % 8 == 0,
                  "mmg_dropout_bf16: bad pointer / shape (M=%lld C=%d)", M, C);
    DropRows a = {rows, (long)M, C, {}};
    if (drop_args("mmg_dropout_bf16", p, seed, site, a.d)) return 1;
    const long work = (long)M * (C / 8);
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3((unsigned)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192)), dim3(256), 0, stream,
                       (const bf16_t*)in, ldi, (bf16_t*)out, ldo, a);
    MMG_LAUNCH_CHECK("mmg_dropout_bf16");
    return 0;
}
