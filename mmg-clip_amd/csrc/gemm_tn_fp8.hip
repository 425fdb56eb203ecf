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
    int xcd_split;      // S of t8_xcd_map
    int xcd_map;        // 1: 1-D grid, all tiles of a row chunk on one XCD (workgroup L runs on XCD L % 8 - observed dispatch; speed only)
};

__device__ __forceinline__ int t8_swz(int m) { return ((m >> 1) & 3) | (((m >> 5) & 1) << 2); }

// Round 4 (after the first C5 profile): with a (tiles, chunks) grid the tiles of one row chunk - which all read the same rows of A and B - were dealt
// round-robin over the 8 XCDs, so every XCD's L2 fetched every row: M (N1 tiles2 + N2 tiles1) bytes through the fabric instead of M (N1 + N2).
// 1-D grid: workgroup L -> XCD L % 8; the XCD's workgroups take (chunk, tile) pairs chunk-major, chunks dealt over the XCDs.
// With fewer than 8 chunks (short reductions: ConvNeXt-B stage 4) a chunk's tiles are split into S groups so that chunks x S is a multiple of 8
// (first version: chunks dealt alone - 4 chunks left half of the XCDs idle and the launch twice as long).
__device__ __forceinline__ void t8_xcd_map(int L, int tiles, int S, int& tile, int& chunk) {
    const int xcd = L & 7, slot = L >> 3, tpu = tiles / S;
    const int unit = (slot / tpu) * 8 + xcd;
    chunk = unit / S;
    tile = (unit % S) * tpu + slot % tpu;
}

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
    int tile = blockIdx.x, chunk = blockIdx.y;
    if (g.xcd_map) t8_xcd_map(blockIdx.x, g.tiles1 * g.tiles2, g.xcd_split, tile, chunk);
    if (chunk >= g.chunks) return;
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


// ---- 256 x 256 tile, 8 waves (round 4) ---------------------------------------------------------------------------------------------------------
// The 128 x 128 kernel above reads half a fragment per MFMA from LDS (64 x 64 wave tiles) and keeps one 32 KB stage in flight per workgroup for ~1 000
// cycles of compute - less than the memory latency under load.  Here a wave owns 128 (A columns) x 64 (B columns): 12 fragments for 32 MFMAs, a stage
// is 64 KB and covers ~2 000 cycles of MFMA work per SIMD, one workgroup per CU.  Rows of 256 bytes start at bank 0, so the swizzle spreads the 16 rows
// a half wave touches in one transposed read (8 q x 2 k-groups) over all 16 slots: slot ^= (m & 7) | (((m >> 5) & 1) << 3).
#define W8_T 256
#define W8_THREADS 512
#define W8_SUB (T8_BK * W8_T)

__device__ __forceinline__ int w8_swz(int m) { return (m & 7) | (((m >> 5) & 1) << 3); }

__device__ __forceinline__ void w8_stage(const unsigned char* __restrict__ G, int ld, int m0, int M, int col0, int ncols, char* lds_tile, int tid) {
#pragma unroll
    for (int it = 0; it < (T8_BK * 16) / W8_THREADS; ++it) {
        const int p = it * W8_THREADS + tid;
        const int r = p >> 4, s = p & 15;
        const int c = s ^ w8_swz(r);
        const int gm = min(m0 + r, M - 1);
        const int gc = max(min(col0 + c * 16, ncols - 16), 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(G + (size_t)gm * ld + gc),
                                         (__attribute__((address_space(3))) void*)(lds_tile + (size_t)(it * W8_THREADS + (tid & ~63)) * 16), 16, 0, 0);
    }
}

// One K = 128 fragment = four transposed reads 8 rows apart (2 048 bytes), issued as INLINE ASSEMBLY: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in
// front of the ds_read_tr builtin while an LDS-DMA is in flight (first build of this kernel: the next stage's eight DMA instructions were drained
// before the first fragment read of every stage - no overlap of memory latency and MFMA work at all; same finding as gemm_tn_wide.hip).  Asm loads are
// invisible to the compiler's counters: the caller waits with w8_lds_wait<N>() (LDS returns in order; no scalar loads inside the stage loop).
// `a` = lane address of the fragment's first row block: stage base + (32 g + q) * 256 + ((cb ^ swz) << 4) + 8 p  (the swizzle of a lane does not
// depend on the row block: 8 r touches neither bits 0-2 nor bit 5 of m).
__device__ __forceinline__ void w8_read_frag(unsigned a, t8_i32x2& r0, t8_i32x2& r1, t8_i32x2& r2, t8_i32x2& r3) {
    asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(r0) : "v"(a));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:2048" : "=v"(r1) : "v"(a));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:4096" : "=v"(r2) : "v"(a));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:6144" : "=v"(r3) : "v"(a));
}
template <int N>
__device__ __forceinline__ void w8_lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
struct W8Frag { t8_i32x2 r0, r1, r2, r3; };
__device__ __forceinline__ i32x8 w8_join(const W8Frag& f) { return i32x8{f.r0[0], f.r0[1], f.r1[0], f.r1[1], f.r2[0], f.r2[1], f.r3[0], f.r3[1]}; }

// CS: this launch also accumulates the column sums of A (32 more accumulator registers in the waves that own B's first 64 columns)
template <int AF, bool CS>
__global__ __launch_bounds__(W8_THREADS, 2) void gemm_tn8_wide_kernel(const GemmTN8 g) {
    constexpr int FM = 8, FN = 4, LDCS = W8_T + 4, STAGE = 2 * W8_SUB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w1 = wave >> 2, w2 = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    int tile, chunk;
    t8_xcd_map(blockIdx.x, g.tiles1 * g.tiles2, g.xcd_split, tile, chunk);
    if (chunk >= g.chunks) return;
    const int t1 = tile / g.tiles2, t2 = tile - t1 * g.tiles2;
    const int c1 = t1 * W8_T, c2 = t2 * W8_T;
    const int m_begin = chunk * g.rows_per_chunk;
    const int m_end = min(m_begin + g.rows_per_chunk, g.M);
    if (m_begin >= m_end) return;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_colsum = CS && (g.colsum_a != nullptr) && (t2 == 0) && (w2 == 0);
    // bias gradient (CS): the column sums of this wave's EIGHT A fragments in ONE accumulator tile - fragment i is multiplied by a selector operand
    // that holds ones (0x38 = 1.0 e4m3) in output row i only, so row i of the tile collects fragment i's 16 column sums (an all-ones operand per
    // fragment, as in the 128 x 128 kernel, costs 8 x 4 registers here and spilled the staging addresses into the stage loop)
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
    const int one4 = 0x38383838;
    // per-lane LDS addressing of the fragments (see w8_read_frag): row part, and the swizzled 16-byte slot of every fragment this wave reads
    const int fq = (lane & 15) >> 1, fp = lane & 1, fsw = fq | ((lg & 1) << 3);
    const unsigned lane_row = (unsigned)((32 * lg + fq) * W8_T + fp * 8);
    unsigned aslot[FM], bslot[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) aslot[i] = (unsigned)(((w1 * FM + i) ^ fsw) << 4);
#pragma unroll
    for (int j = 0; j < FN; ++j) bslot[j] = (unsigned)(((w2 * FN + j) ^ fsw) << 4);

    auto stage = [&](int buf, int m0) {
        char* base = smem + buf * STAGE;
        w8_stage(g.A, g.lda, m0, g.M, c1, g.N1, base, tid);
        w8_stage(g.B, g.ldb, m0, g.M, c2, g.N2, base + W8_SUB, tid);
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
            const int valid = m_end - mt, per = (T8_BK - valid) * 16;
            for (int p = tid; p < per * 2; p += W8_THREADS) {
                const int img = p / per, q = p - img * per;
                *reinterpret_cast<uint4*>(cbase + img * W8_SUB + (valid + (q >> 4)) * W8_T + (q & 15) * 16) = make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
        }
        if (kt + 1 < nk) stage(cur ^ 1, mt + T8_BK);
        // fragments: B's four first, then A's eight one ahead of their MFMAs in two register sets; counted lgkmcnt (4 reads per fragment)
        const unsigned sbase = (unsigned)(cur * STAGE) + lane_row;
        W8Frag fb[FN], fa[2];
#pragma unroll
        for (int j = 0; j < FN; ++j) w8_read_frag(sbase + W8_SUB + bslot[j], fb[j].r0, fb[j].r1, fb[j].r2, fb[j].r3);
        w8_read_frag(sbase + aslot[0], fa[0].r0, fa[0].r1, fa[0].r2, fa[0].r3);
        i32x8 bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if (i + 1 < FM) {
                w8_read_frag(sbase + aslot[i + 1], fa[(i + 1) & 1].r0, fa[(i + 1) & 1].r1, fa[(i + 1) & 1].r2, fa[(i + 1) & 1].r3);
                w8_lds_wait<4>();
            } else {
                w8_lds_wait<0>();
            }
            if (i == 0) {
#pragma unroll
                for (int j = 0; j < FN; ++j) bfr[j] = w8_join(fb[j]);
            }
            const i32x8 af = w8_join(fa[i & 1]);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af, acc[i][j], 0, AF, 0, 0, 0, 0);
            if (CS && do_colsum) {
                int sv = li == i ? one4 : 0;
                asm volatile("" : "+v"(sv));          // opaque: hipcc otherwise keeps all eight selectors (64 registers) live across the stage loop
                const i32x8 sel = {sv, sv, sv, sv, sv, sv, sv, sv};
                accb = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(sel, af, accb, 0, AF, 0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);        // (the next fragment's reads stay behind this fragment's MFMAs: its register set is the one read two steps ago)
        }
    }
    const float alpha = g.alpha_dev ? g.alpha * *g.alpha_dev : g.alpha;
    if (CS && do_colsum && lg < FM / 4) {          // lane (li, lg) holds tile element [column li of the fragment][selector row 4 lg + e]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n1 = c1 + (w1 * FM + 4 * lg + e) * 16 + li;
            if (n1 < g.N1) atomicAdd(g.colsum_a + n1, accb[e] * alpha);
        }
    }
    // lane (li, lg) holds C[n1 = c1 + 128 w1 + 16 i + li][n2 = c2 + 64 w2 + 16 j + 4 lg + 0..3]; 64-row slabs through LDS, 256-byte contiguous atomic rows
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        __syncthreads();
        if ((sl >> 1) == w1) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int i = (sl & 1) * 4 + i4;
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *reinterpret_cast<f32x4*>(Cs + (i4 * 16 + li) * LDCS + (w2 * FN + j) * 16 + 4 * lg) = acc[i][j];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * W8_T; idx += W8_THREADS) {
            const int r = idx / W8_T, c = idx - r * W8_T;
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
    // 256 x 256 tiles (one 8-wave workgroup per CU) where both widths fill them and the reduction is long; MMG_TN8_WIDE=0 / 1 forces
    const char* ew = getenv("MMG_TN8_WIDE");
    const bool wide = ew ? atoi(ew) != 0 : (N1 >= 256 && N2 >= 256 && M >= 8192);
    const char* ex = getenv("MMG_TN8_XCD");
    g.xcd_map = wide || !(ex && atoi(ex) == 0);
    const int T = wide ? W8_T : T8_T;
    g.tiles1 = cdiv(N1, T); g.tiles2 = cdiv(N2, T);
    const int tiles = g.tiles1 * g.tiles2;
    static const int target_wgs = getenv("MMG_TN8_WGS") ? atoi(getenv("MMG_TN8_WGS")) : 1024;      // two workgroups per CU, two rounds
    int chunks = (wide ? mmg_cu_count_cached() : target_wgs) / tiles;
    if (chunks < 1) chunks = 1;
    const int max_chunks = cdiv(M, T8_BK);
    if (chunks > max_chunks) chunks = max_chunks;
    g.rows_per_chunk = cdiv(cdiv(M, chunks), T8_BK) * T8_BK;
    g.chunks = cdiv(M, g.rows_per_chunk);
    int S = 1;                                                 // (see t8_xcd_map)
    while (S < 8 && ((g.chunks * S) % 8 != 0) && tiles % (2 * S) == 0) S *= 2;
    g.xcd_split = S;
    const int units8 = cdiv(g.chunks * S, 8) * 8;              // units are dealt in groups of 8 (workgroups past the last chunk return at once)
    const dim3 grid = g.xcd_map ? dim3((tiles / S) * units8) : dim3(tiles, g.chunks);
    if (wide) {
        const size_t shm = 2 * (size_t)(2 * W8_SUB);           // (the fp32 flush slab, 64 x 260 x 4, fits inside)
        MMG_NOTE_KERNEL("gemm_tn8_wide_kernel<%d>", a_e5m2 ? 1 : 0);
#define W8_LAUNCH(AF_, CS_) do { mmg_allow_lds(gemm_tn8_wide_kernel<AF_, CS_>, shm); \
                                 hipLaunchKernelGGL((gemm_tn8_wide_kernel<AF_, CS_>), grid, dim3(W8_THREADS), shm, stream, g); } while (0)
        if (a_e5m2) { if (colsum_a) W8_LAUNCH(1, true); else W8_LAUNCH(1, false); }
        else { if (colsum_a) W8_LAUNCH(0, true); else W8_LAUNCH(0, false); }
#undef W8_LAUNCH
        MMG_LAUNCH_CHECK("mmg_gemm_tn_fp8");
        return 0;
    }
    const size_t stage = 2 * (size_t)(2 * T8_SUB), cs = (size_t)64 * (T8_T + 4) * 4;
    const size_t shm = stage > cs ? stage : cs;
    MMG_NOTE_KERNEL("gemm_tn8_kernel<%d>", a_e5m2 ? 1 : 0);
    if (a_e5m2) {
        mmg_allow_lds(gemm_tn8_kernel<1>, shm);
        hipLaunchKernelGGL(gemm_tn8_kernel<1>, grid, dim3(T8_THREADS), shm, stream, g);
    } else {
        mmg_allow_lds(gemm_tn8_kernel<0>, shm);
        hipLaunchKernelGGL(gemm_tn8_kernel<0>, grid, dim3(T8_THREADS), shm, stream, g);
    }
    MMG_LAUNCH_CHECK("mmg_gemm_tn_fp8");
    return 0;
}
