// 8-bit weight-gradient GEMM for the fp8 ConvNeXt path (BASELINE config C5, "ConvNeXt-base fp8 MFMA path"; round 4):
//     C[N1,N2] (fp32) += alpha * alpha_dev * A[M,N1]^T B[M,N2],      A: OCP e5m2 (gradients) or e4m3 bytes, B: OCP e4m3 bytes (activations)
// The reference has no fp8 (its encoders run in fp32: mmgclip/networks/encoder.py:53,138); this is the weight gradient of torchvision CNBlock's two
// Linear layers (dW2 = dy^T GELU(h), dW1 = dh^T LN(x)) on the operands the fp8 forward / data-gradient GEMMs of this library already hold in 8 bits,
// so that `dh` is written ONCE, in 8 bits, and nothing is re-quantised.
//
// The reduction index (the row m) is the slow index of both operands, and v_mfma_f32_16x16x128_f8f6f4 wants 32 consecutive k per lane: the fragments are
// read with ds_read_b64_tr_b8 - per 16 lanes a block of 8 rows x 16 byte-columns, lane 2q + p supplying the address of row q, bytes 8p .. 8p + 7, lane i
// receiving column i of the 8 rows (probed: tools/micro/tr_b8_probe.hip, profiles/r04_tr_b8_probe.txt) - four reads per fragment.  A and B are read by the
// same rule, so whatever order the MFMA gives the 32 bytes of a lane, both operands present the same k in the same byte.
// One workgroup (4 waves, 2 x 2) owns a 128 x 128 output tile for a chunk of rows; stages of 128 rows (one MFMA k-step: 16 KiB per operand) arrive by
// 16-byte global_load_lds into a double buffer; the LDS image is lane-linear, rows of 128 bytes, and the 16-byte slot of a row is XOR-ed (on the source
// address and on the read) with ((m >> 1) & 3) | (((m >> 5) & 1) << 2): the 8 rows of a transposed read then sit in 8 different (slot, bank-row half)
// places and the two 16-lane groups of a half wave (k ranges 32 apart) in disjoint slots - conflict-free by the bank model of the microarchitecture guide.
#include "common.h"
#include <stdlib.h>

#define T8_T 128          // output tile edge
#define T8_BK 128         // reduction rows per stage = one k-step of the K = 128 MFMA
#define T8_THREADS 256
#define T8_SUB (T8_BK * T8_T)          // bytes of one operand's stage image

struct GemmTN8 {
    const unsigned char* A; const unsigned char* B;
    int M, N1, N2, lda, ldb;
    float* C; int ldc;
    float* colsum_a;
    float alpha; const float* alpha_dev;
    int tiles1, tiles2, rows_per_chunk, chunks;
};

__device__ __forceinline__ int t8_swz(int m) { return ((m >> 1) & 3) | (((m >> 5) & 1) << 2); }

// rows m0 .. m0 + 127 (clamped to M - 1) x 128 byte-columns col0 .. (clamped) -> lane-linear LDS image, swizzled on the source side
__device__ __forceinline__ void t8_stage(const unsigned char* __restrict__ G, int ld, int m0, int M, int col0, int ncols, char* lds_tile, int tid) {
#pragma unroll
    for (int it = 0; it < (T8_BK * 8) / T8_THREADS; ++it) {
        const int p = it * T8_THREADS + tid;
        const int r = p >> 3, s = p & 7;
        const int c = s ^ t8_swz(r);                       // logical 16-byte chunk stored at physical slot s
        const int gm = min(m0 + r, M - 1);
        const int gc = max(min(col0 + c * 16, ncols - 16), 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(G + (size_t)gm * ld + gc),
                                         (__attribute__((address_space(3))) void*)(lds_tile + (size_t)(it * T8_THREADS + (tid & ~63)) * 16), 16, 0, 0);
    }
}

// fragment of the 16-column block cb (0..7) of a stage image: lane (i = lane & 15, g = lane >> 4) receives, for k = 32 g .. 32 g + 31, tile[k][16 cb + i]
typedef __attribute__((ext_vector_type(2))) int t8_i32x2;
typedef __attribute__((address_space(3))) t8_i32x2 t8_lds_v2;
__device__ __forceinline__ i32x8 t8_frag(const char* lds_tile, int cb, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 1, p = lane & 1;
    i32x8 f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 32 * g + 8 * r + q;
        const t8_i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((t8_lds_v2*)(lds_tile + m * T8_T + ((cb ^ t8_swz(m)) << 4) + p * 8));
        f[2 * r] = v[0]; f[2 * r + 1] = v[1];
    }
    return f;
}

// AF: format of A - 0 = e4m3, 1 = e5m2 (the f8f6f4 MFMA's format codes); B is e4m3
template <int AF>
__global__ __launch_bounds__(T8_THREADS, 2) void gemm_tn8_kernel(const GemmTN8 g) {
    constexpr int FM = 4, FN = 4, LDCS = T8_T + 4, STAGE = 2 * T8_SUB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int w1 = wave >> 1, w2 = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x, chunk = blockIdx.y;
    const int t1 = tile / g.tiles2, t2 = tile - t1 * g.tiles2;
    const int c1 = t1 * T8_T, c2 = t2 * T8_T;
    const int m_begin = chunk * g.rows_per_chunk;
    const int m_end = min(m_begin + g.rows_per_chunk, g.M);
    if (m_begin >= m_end) return;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient: one extra MFMA per A fragment against an all-ones e4m3 operand gives the column sums of A in every accumulator row
    const bool do_colsum = (g.colsum_a != nullptr) && (t2 == 0) && (w2 == 0);
    f32x4 accb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int one4 = 0x38383838;                                   // four e4m3 ones
    const i32x8 ones = {one4, one4, one4, one4, one4, one4, one4, one4};

    auto stage = [&](int buf, int m0) {
        char* base = smem + buf * STAGE;
        t8_stage(g.A, g.lda, m0, g.M, c1, g.N1, base, tid);
        t8_stage(g.B, g.ldb, m0, g.M, c2, g.N2, base + T8_SUB, tid);
    };
    const int nk = (m_end - m_begin + T8_BK - 1) / T8_BK;
    stage(0, m_begin);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        const int mt = m_begin + kt * T8_BK;
        char* cbase = smem + cur * STAGE;
        if (mt + T8_BK > m_end) {
            // ragged end of the reduction: rows >= m_end must contribute zero (they hold clamped duplicates); a zero byte is 0.0 in both formats
            const int valid = m_end - mt, per = (T8_BK - valid) * 8;
            for (int p = tid; p < per * 2; p += T8_THREADS) {
                const int img = p / per, q = p - img * per;
                *reinterpret_cast<uint4*>(cbase + img * T8_SUB + (valid + (q >> 3)) * T8_T + (q & 7) * 16) = make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
        }
        if (kt + 1 < nk) stage(cur ^ 1, mt + T8_BK);
        i32x8 af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = t8_frag(cbase, w1 * FM + i, lane);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = t8_frag(cbase + T8_SUB, w2 * FN + j, lane);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, AF, 0, 0, 0, 0);
        if (do_colsum) {
#pragma unroll
            for (int i = 0; i < FM; ++i) accb[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, af[i], accb[i], 0, AF, 0, 0, 0, 0);
        }
    }
    const float alpha = g.alpha_dev ? g.alpha * *g.alpha_dev : g.alpha;
    if (do_colsum && lg == 0) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int n1 = c1 + (w1 * FM + i) * 16 + li;
            if (n1 < g.N1) atomicAdd(g.colsum_a + n1, accb[i][0] * alpha);
        }
    }
    // swapped issue: lane (li, lg) holds C[n1 = .. + li][n2 = .. + 4 lg + 0..3]; 64-row slabs go through LDS and leave as 256-byte contiguous atomic rows
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        __syncthreads();
        if (sl == w1) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *reinterpret_cast<f32x4*>(Cs + (i * 16 + li) * LDCS + (w2 * FN + j) * 16 + 4 * lg) = acc[i][j];
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * T8_T; idx += T8_THREADS) {
            const int r = idx / T8_T, c = idx - r * T8_T;
            const int gr = c1 + sl * 64 + r, gc = c2 + c;
            if (gr < g.N1 && gc < g.N2) atomicAdd(g.C + (size_t)gr * g.ldc + gc, Cs[r * LDCS + c] * alpha);
        }
    }
}

MMG_API int mmg_gemm_tn_fp8(const void* A, int lda, int a_e5m2, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                            float alpha, const float* alpha_dev, float* colsum_a, hipStream_t stream) {
    MMG_CHECK_ARG(A && B && C, "mmg_gemm_tn_fp8: null operand");
    MMG_CHECK_ARG(M > 0 && N1 >= 16 && N2 >= 16, "mmg_gemm_tn_fp8: M=%d N1=%d N2=%d", M, N1, N2);
    MMG_CHECK_ARG(N1 % 16 == 0 && N2 % 16 == 0 && lda % 16 == 0 && ldb % 16 == 0 && lda >= N1 && ldb >= N2 && ldc >= N2,
                  "mmg_gemm_tn_fp8: N1=%d N2=%d lda=%d ldb=%d ldc=%d must be multiples of 16 (bytes) and consistent", N1, N2, lda, ldb, ldc);
    GemmTN8 g;
    g.A = (const unsigned char*)A; g.B = (const unsigned char*)B; g.M = M; g.N1 = N1; g.N2 = N2; g.lda = lda; g.ldb = ldb;
    g.C = C; g.ldc = ldc; g.colsum_a = colsum_a; g.alpha = alpha; g.alpha_dev = alpha_dev;
    g.tiles1 = cdiv(N1, T8_T); g.tiles2 = cdiv(N2, T8_T);
    const int tiles = g.tiles1 * g.tiles2;
    static const int target_wgs = getenv("MMG_TN8_WGS") ? atoi(getenv("MMG_TN8_WGS")) : 1024;      // two workgroups per CU, two rounds
    int chunks = target_wgs / tiles;
    if (chunks < 1) chunks = 1;
    const int max_chunks = cdiv(M, T8_BK);
    if (chunks > max_chunks) chunks = max_chunks;
    g.rows_per_chunk = cdiv(cdiv(M, chunks), T8_BK) * T8_BK;
    g.chunks = cdiv(M, g.rows_per_chunk);
    const size_t stage = 2 * (size_t)(2 * T8_SUB), cs = (size_t)64 * (T8_T + 4) * 4;
    const size_t shm = stage > cs ? stage : cs;
    MMG_NOTE_KERNEL("gemm_tn8_kernel<%d>", a_e5m2 ? 1 : 0);
    if (a_e5m2) {
        mmg_allow_lds(gemm_tn8_kernel<1>, shm);
        hipLaunchKernelGGL(gemm_tn8_kernel<1>, dim3(tiles, g.chunks), dim3(T8_THREADS), shm, stream, g);
    } else {
        mmg_allow_lds(gemm_tn8_kernel<0>, shm);
        hipLaunchKernelGGL(gemm_tn8_kernel<0>, dim3(tiles, g.chunks), dim3(T8_THREADS), shm, stream, g);
    }
    MMG_LAUNCH_CHECK("mmg_gemm_tn_fp8");
    return 0;
}
