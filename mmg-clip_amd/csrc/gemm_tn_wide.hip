// Weight-gradient GEMM for long reductions:  C[N1,N2] += alpha * A[M,N1]^T B[M,N2]   (bf16 in, fp32 atomics out), M >> N1, N2.
//
// Replaces, like gemm_tn_kernel (gemm_bf16.hip), the autograd backward of nn.Linear with respect to its weight in the towers
// (torchvision CNBlock Linear(C,4C) / Linear(4C,C), the 2x2 downsampling convolutions: mmgclip/networks/encoder.py:53 and the
// module tree of notebooks/clf_convnext_tiny_experimental.ipynb cell 3).  On the ConvNeXt shapes the reduction runs over
// M = 0.26...16.8 million pixel rows onto at most 768 x 3072 outputs, so the kernel is a STREAM of the two operands:
//
//   * one 8-wave workgroup per CU owns a whole 192 x 384 (or 384 x 192 / 96 x 384 / 384 x 96) fp32 accumulator tile - 144
//     registers per lane - for one contiguous chunk of rows, so every operand byte is staged 1/(tiles across) times instead of
//     once per 128-wide tile (gemm_tn_kernel: 2.4...3.6x the algorithmic bytes through L2, 1.9x through HBM);
//   * rows arrive by LDS-DMA (16-byte global_load_lds) into a ring of four 32-row stages, three of them in flight: counted
//     s_waitcnt vmcnt + ONE raw s_barrier per stage (a __syncthreads() would drain the DMA queue);
//   * the LDS image is cut into 64-column panels of 128-byte rows, lane-linear as LDS-DMA requires, with the XOR swizzle on the
//     SOURCE address; both operands are reduction-major, so both MFMA fragments come from ds_read_b64_tr_b16 (hardware
//     transpose), conflict-free: a 32-lane half reads 8 rows x 32 bytes that the swizzle spreads over the 8 32-byte slots of a
//     256-byte bank row;
//   * all tiles of a row chunk run on one XCD (blockIdx % 8), so re-reads of a chunk by the other tiles hit that XCD's L2;
//   * accumulators leave through LDS as 256-byte contiguous fp32 atomic rows (the full-rate atomic shape), once per workgroup.
#include "gemm_tn.h"
#include <stdlib.h>
#include <utility>

#define TW_THREADS 512
#define TW_BK 32                      // reduction rows per stage = one 16x16x32 MFMA k-step
#define TW_NS 4                       // ring slots
#define TW_IMG (TW_BK * 128)          // bytes of one 64-column panel of a stage

template <int T1, int T2> struct TwCfg;
template <> struct TwCfg<192, 384> { static constexpr int W1 = 2, W2 = 4, FM = 6, FN = 6; };
template <> struct TwCfg<384, 192> { static constexpr int W1 = 4, W2 = 2, FM = 6, FN = 6; };
template <> struct TwCfg<96, 384>  { static constexpr int W1 = 1, W2 = 8, FM = 6, FN = 3; };
template <> struct TwCfg<384, 96>  { static constexpr int W1 = 8, W2 = 1, FM = 3, FN = 6; };
// ConvNeXt-B widths (128 / 256 / 512 / 1024): 256-wide tiles, 64 / 128 accumulator registers per lane
template <> struct TwCfg<128, 256> { static constexpr int W1 = 2, W2 = 4, FM = 4, FN = 4; };
template <> struct TwCfg<256, 256> { static constexpr int W1 = 2, W2 = 4, FM = 8, FN = 4; };

// physical 16-byte slot (0..7) of logical chunk c in reduction row m of a panel
__device__ __forceinline__ int tw_swz(int m, int c) {
    const int f = ((m >> 1) & 1) | (((m >> 3) & 1) << 1);
    return (((c >> 1) ^ f) << 1) | (c & 1);
}

__device__ __forceinline__ void tw_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(n) with n = loads per stage of this wave x stages allowed to stay in flight
template <int L>
__device__ __forceinline__ void tw_wait(int inflight) {
    if (inflight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
    else if (inflight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// transposed fragment: lane (i = lane & 15, g = lane >> 4) receives panel[8g + 0..7][cb*16 + i]; `a` = this lane's LDS byte
// address of its first 4-row block (rows 8g + q), the second block (rows + 4) has the same swizzle and sits 512 bytes further.
// Issued as inline assembly ON PURPOSE: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr builtin whenever an
// LDS-DMA is in flight (it cannot tell the ring slots apart), which drains the three stages this kernel keeps in flight - plain
// ds_read_b128 loads do not get that wait, the transposed-read builtin does.  An asm load is invisible to the compiler's
// counters, so the reads of a stage are followed by tw_lds_wait<N>() + a scheduling fence before the first MFMA that uses them.
template <int OFF>
__device__ __forceinline__ void tw_read_frag(unsigned a, bf16x4& lo, bf16x4& hi) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 512));
}
template <int N>
__device__ __forceinline__ void tw_lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);        // no MFMA may be scheduled above the wait ("memory" does not order register-only code)
}
// c + (sum of the two bf16 of d).  The empty asm makes the dword opaque: hipcc (ROCm 7.2) otherwise folds "element e of a vector,
// reinterpreted as a bf16 pair" to element 0 for every e (v_dot2c reads the same register four times).
__device__ __forceinline__ float tw_dot2_ones(int d, float c) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    asm volatile("" : "+v"(d));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, d), __builtin_bit_cast(bf2_t, 0x3F803F80u), c, false);
}
__device__ __forceinline__ bf16x8 tw_join(const bf16x4 lo, const bf16x4 hi) {
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>) (fragment indices must be constants: they select immediates)
template <int... Is, class F>
__device__ __forceinline__ void tw_static_for(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
#define TW_FOR(N, I, ...) tw_static_for(std::make_integer_sequence<int, N>{}, [&](auto I##_c) { constexpr int I = decltype(I##_c)::value; __VA_ARGS__ })

template <int T1, int T2>
__global__ __launch_bounds__(TW_THREADS, 2) void gemm_tn_wide_kernel(const GemmTN g) {
    using Cfg = TwCfg<T1, T2>;
    constexpr int W1 = Cfg::W1, W2 = Cfg::W2, FM = Cfg::FM, FN = Cfg::FN;
    constexpr int NIA = (T1 + 63) / 64, NIB = (T2 + 63) / 64;          // 64-column panels per stage
    constexpr int STAGE = (NIA + NIB) * TW_IMG;
    constexpr int TOT = (NIA + NIB) * 4;                                // 1-KiB LDS-DMA instructions per stage
    constexpr int SLOTS = (TOT + 7) / 8;                                // per wave (the last slot only on the first waves)
    constexpr int NHI = TOT - 8 * (SLOTS - 1);                          // waves that issue SLOTS instructions; the others SLOTS-1
    constexpr int LDCS = T2 + 4;
    static_assert(W1 * W2 == 8 && W1 * FM * 16 == T1 && W2 * FN * 16 == T2, "wave grid must cover the tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w1 = wave / W2, w2 = wave - w1 * W2;
    const int li = lane & 15, lg = lane >> 4;

    // workgroup -> (row chunk, tile): the tiles of one chunk sit on one XCD (blockIdx % 8) and run side by side
    const int ntile = g.tiles1 * g.tiles2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunk = (slot / ntile) * 8 + xcd, tile = slot % ntile;
    const int t1 = tile / g.tiles2, t2 = tile - t1 * g.tiles2;
    const int c1 = t1 * T1, c2 = t2 * T2;
    const int m_begin = chunk * g.rows_per_chunk;                       // (multiple of 32)
    const int m_end = min(m_begin + g.rows_per_chunk, g.M);
    if (m_begin >= m_end) return;                                       // whole workgroup leaves: no barrier is pending
    const int nk = (m_end - m_begin + TW_BK - 1) / TW_BK;

    // ---- staging plan of this wave: slot k issues instruction id = 8k + wave = (panel, 8-row group) ------------------------
    const bf16_t* src[SLOTS];                 // wave-uniform operand base (A or B) at row m_begin
    unsigned goff[SLOTS];                     // per-lane byte offset inside a 32-row stage (row * ld + swizzled column)
    int ldst[SLOTS];                          // wave-uniform LDS byte offset inside a stage
    long rstep[SLOTS];                        // bytes per stage step of that operand
    // slot k -> (operand, row r of the stage, first column, leading dimension); columns beyond the matrix (tile wider than N)
    // are clamped onto valid ones: the instruction count per wave stays a compile-time constant (counted vmcnt), the
    // duplicates cost L2 hits only and feed accumulators that are never flushed
    auto slot_geometry = [&](int k, bool& isA, int& r, int& col, int& ld, int& dst) {
        const int id = 8 * k + wave;
        const int img = min(id >> 2, NIA + NIB - 1), t = id & 3;
        isA = img < NIA;
        r = 8 * t + (lane >> 3);
        const int c = tw_swz(r, lane & 7);
        ld = isA ? g.lda : g.ldb;
        col = min((isA ? c1 + img * 64 : c2 + (img - NIA) * 64) + c * 8, (isA ? g.N1 : g.N2) - 8);
        dst = img * TW_IMG + t * 1024;
    };
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        bool isA; int r, col, ld;
        slot_geometry(k, isA, r, col, ld, ldst[k]);
        src[k] = (isA ? g.A : g.B) + (size_t)m_begin * ld;
        goff[k] = (unsigned)(r * ld + col) * 2u;
        rstep[k] = (long)TW_BK * ld * 2;
    }
    const bool hi = wave < NHI;               // this wave also issues the last slot

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            if (k + 1 < SLOTS || hi)
                tw_glds16(reinterpret_cast<const char*>(src[k]) + (size_t)kt * rstep[k] + goff[k], base + ldst[k]);
        }
    };
    // last stage of the LAST chunk when M is not a multiple of 32: rows >= M are clamped onto row M-1 (valid memory) here and
    // zeroed in the A panels after they have landed
    auto stage_ragged = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        const int m0 = m_begin + kt * TW_BK;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            if (k + 1 < SLOTS || hi) {
                bool isA; int r, col, ld, dst;
                slot_geometry(k, isA, r, col, ld, dst);
                const int gm = min(m0 + r, g.M - 1) - m_begin;
                tw_glds16(reinterpret_cast<const char*>(src[k]) + ((size_t)gm * ld + col) * 2, base + dst);
            }
        }
    };
    const bool ragged = (m_end - m_begin) % TW_BK != 0;
    auto issue = [&](int buf, int kt) {
        if (ragged && kt == nk - 1) stage_ragged(buf, kt); else stage(buf, kt);
    };

    // ---- fragment addresses -------------------------------------------------------------------------------------------------
    // The wave grid is INTERLEAVED over the tile: fragment i of wave row w1 is the 16-column block i*W1 + w1 of the A tile (and
    // j*W2 + w2 of the B tile).  Panel (block >> 2) and position in the panel (block & 3) then split into a compile-time part -
    // an immediate of the ds_read - and a per-wave part that needs 4 / W base registers (1 for a 4-wide wave grid, 2 for 2, 4 for 1)
    // instead of one address register per fragment.
    constexpr int NVA = W1 >= 4 ? 1 : 4 / W1, NVB = W2 >= 4 ? 1 : 4 / W2;
    const int q = (lane >> 2) & 3, p = lane & 3;
    const int m_a = 8 * lg + q;
    const int fsw = ((m_a >> 1) & 1) | (((m_a >> 3) & 1) << 1);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;    // LDS byte address of the ring
    unsigned baseA[NVA], baseB[NVB];
#pragma unroll
    for (int v = 0; v < NVA; ++v)
        baseA[v] = lds0 + (w1 >> 2) * TW_IMG + m_a * 128 + ((((v * W1 + w1) & 3) ^ fsw) << 5) + p * 8;
#pragma unroll
    for (int v = 0; v < NVB; ++v)
        baseB[v] = lds0 + (NIA + (w2 >> 2)) * TW_IMG + m_a * 128 + ((((v * W2 + w2) & 3) ^ fsw) << 5) + p * 8;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of A, in the first tile column only: the lane adds up its own 8 reduction rows of each A
    // fragment with v_dot2c_f32_bf16 against (1, 1) - one register per fragment (an MFMA against an all-ones operand would keep
    // four), the VALU is idle in this kernel; the 4 lane groups of a column are summed at the end
    const bool do_colsum = (g.colsum_a != nullptr) && !g.swapped && (t2 == 0) && (w2 == 0);
    float csum[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) csum[i] = 0.f;
    // exchanged operands (g.swapped): the caller's A is this kernel's B - the same sums over the B fragments, first tile ROW only
    const bool do_colsum_b = (g.colsum_a != nullptr) && g.swapped && (t1 == 0) && (w1 == 0);
    float csum_b[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) csum_b[j] = 0.f;

#pragma unroll
    for (int s = 0; s < TW_NS - 1; ++s)
        if (s < nk) issue(s, s);

    for (int kt0 = 0; kt0 < nk; kt0 += TW_NS) {
#pragma unroll
        for (int s = 0; s < TW_NS; ++s) {
            const int kt = kt0 + s;
            if (kt < nk) {
                // stage kt has landed once only the (up to two) younger stages of this wave are outstanding
                const int inflight = min(TW_NS - 2, nk - 1 - kt);
                if (hi) tw_wait<SLOTS>(inflight); else tw_wait<SLOTS - 1>(inflight);
                __builtin_amdgcn_s_barrier();          // every wave's part of stage kt is in LDS; stage kt-1 has been consumed
                if (ragged && kt == nk - 1) {
                    const int valid = (m_end - m_begin) - kt * TW_BK;          // rows of this stage inside the matrix
                    // (A panels: the products vanish; with exchanged operands the column sums run over the B panels: those too)
                    const int nimg = g.swapped ? NIA + NIB : NIA;
                    for (int e = tid; e < nimg * (TW_BK - valid) * 8; e += TW_THREADS) {
                        const int img = e / ((TW_BK - valid) * 8), rem = e - img * ((TW_BK - valid) * 8);
                        *reinterpret_cast<uint4*>(smem + s * STAGE + img * TW_IMG + (valid + (rem >> 3)) * 128 + (rem & 7) * 16) =
                            make_uint4(0, 0, 0, 0);
                    }
                    __syncthreads();                   // (nothing is in flight any more in the last stage)
                }
                if (kt + TW_NS - 1 < nk) issue((s + TW_NS - 1) % TW_NS, kt + TW_NS - 1);
                // A fragments and the first two B fragments are requested up front; then, per B fragment j: wait for it (LDS
                // returns a wave's reads in order: at most the one younger fragment may still be out), issue its FM MFMAs, request
                // fragment j + 2.  Three B fragments rotate through registers instead of FN.
                bf16x4 alo[FM], ahi[FM], blo[3], bhi[3];
                unsigned ra[NVA], rb[NVB];
#pragma unroll
                for (int v = 0; v < NVA; ++v) ra[v] = baseA[v] + s * STAGE;
#pragma unroll
                for (int v = 0; v < NVB; ++v) rb[v] = baseB[v] + s * STAGE;
                TW_FOR(FM, i, tw_read_frag<((i * W1) >> 2) * TW_IMG>(ra[i % NVA], alo[i], ahi[i]););
                TW_FOR(2, j, tw_read_frag<((j * W2) >> 2) * TW_IMG>(rb[j % NVB], blo[j], bhi[j]););
                bf16x8 af[FM];
                TW_FOR(FN, j,
                    if (j + 1 < FN) tw_lds_wait<2>(); else tw_lds_wait<0>();
                    if (j == 0) {
                        TW_FOR(FM, i, af[i] = tw_join(alo[i], ahi[i]););
                    }
                    const bf16x8 bj = tw_join(blo[j % 3], bhi[j % 3]);
                    TW_FOR(FM, i, acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bj, af[i], acc[i][j], 0, 0, 0););
                    if (do_colsum_b) {
                        const i32x4 wb = __builtin_bit_cast(i32x4, bj);
                        float cb = csum_b[j];
                        cb = tw_dot2_ones(wb[0], cb); cb = tw_dot2_ones(wb[1], cb); cb = tw_dot2_ones(wb[2], cb); cb = tw_dot2_ones(wb[3], cb);
                        csum_b[j] = cb;
                    }
                    if (j + 2 < FN) tw_read_frag<(((j + 2) * W2) >> 2) * TW_IMG>(rb[(j + 2) % NVB], blo[(j + 2) % 3], bhi[(j + 2) % 3]);
                );
                if (do_colsum) {
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const i32x4 w = __builtin_bit_cast(i32x4, af[i]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) csum[i] = tw_dot2_ones(w[e], csum[i]);
                    }
                }
            }
        }
    }

    if (do_colsum) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float v = csum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n1 = c1 + (i * W1 + w1) * 16 + li;
            if (lg == 0 && n1 < g.N1) atomicAdd(g.colsum_a + n1, v * g.alpha);
        }
    }
    if (do_colsum_b) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            float v = csum_b[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n2 = c2 + (j * W2 + w2) * 16 + li;
            if (lg == 0 && n2 < g.N2) atomicAdd(g.colsum_a + n2, v * g.alpha);
        }
    }
    // swapped issue: lane (li, lg) holds C[n1 = .. + li][n2 = .. + 4 lg + 0..3]; the FM*16 tile rows of one wave row (16-row blocks
    // i*W1 + w1) at a time go through LDS and leave as 256-byte contiguous atomic rows
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll 1
    for (int sl = 0; sl < W1; ++sl) {
        __syncthreads();
        if (w1 == sl) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    *reinterpret_cast<f32x4*>(Cs + (i * 16 + li) * LDCS + (j * W2 + w2) * 16 + 4 * lg) = acc[i][j];
        }
        __syncthreads();
        if (!g.swapped) {
            for (int idx = tid; idx < FM * 16 * T2; idx += TW_THREADS) {
                const int r = idx / T2, c = idx - r * T2;
                const int gr = c1 + ((r >> 4) * W1 + sl) * 16 + (r & 15), gc = c2 + c;
                if (gr < g.N1 && gc < g.N2) atomicAdd(g.C + (size_t)gr * g.ldc + gc, Cs[r * LDCS + c] * g.alpha);
            }
        } else {                 // the caller's matrix is [N2, N1]: 16 consecutive rows of the staged block are 64 contiguous bytes of it
            for (int idx = tid; idx < FM * 16 * T2; idx += TW_THREADS) {
                const int c = idx / (FM * 16), r = idx - c * (FM * 16);
                const int gr = c1 + ((r >> 4) * W1 + sl) * 16 + (r & 15), gc = c2 + c;
                if (gr < g.N1 && gc < g.N2) atomicAdd(g.C + (size_t)gc * g.ldc + gr, Cs[r * LDCS + c] * g.alpha);
            }
        }
    }
}

template <int T1, int T2>
static void launch_tw(GemmTN& g, hipStream_t stream) {
    using Cfg = TwCfg<T1, T2>;
    g.tiles1 = cdiv(g.N1, T1);
    g.tiles2 = cdiv(g.N2, T2);
    const int ntile = g.tiles1 * g.tiles2;
    // one workgroup per CU (149 KiB of LDS): 32 per XCD = the tiles of `cpx` row chunks
    int cpx = 32 / ntile;
    if (cpx < 1) cpx = 1;
    int chunks = 8 * cpx;
    g.rows_per_chunk = cdiv(cdiv(g.M, chunks), TW_BK) * TW_BK;
    g.chunks = chunks;
    constexpr int NIA = (T1 + 63) / 64, NIB = (T2 + 63) / 64;
    const size_t ring = (size_t)TW_NS * (NIA + NIB) * TW_IMG;
    const size_t cs = (size_t)Cfg::FM * 16 * (T2 + 4) * 4;
    const size_t shm = ring > cs ? ring : cs;
    mmg_allow_lds(gemm_tn_wide_kernel<T1, T2>, shm);
    MMG_NOTE_KERNEL("gemm_tn_wide_kernel<%d, %d>", T1, T2);
    hipLaunchKernelGGL((gemm_tn_wide_kernel<T1, T2>), dim3(8 * cpx * ntile), dim3(TW_THREADS), shm, stream, g);
}

bool mmg_tn_wide_launch(GemmTN& g, hipStream_t stream) {
    static const int min_m = getenv("MMG_TN_WIDE8_MIN_M") ? atoi(getenv("MMG_TN_WIDE8_MIN_M")) : 65536;
    // every workgroup flushes a whole tile with fp32 atomics (75 MB per launch at 256 workgroups of 192 x 384 = ~60 us): only
    // reductions long enough to amortise that take this kernel
    if (g.M < min_m || g.N1 < 96 || g.N2 < 96) return false;
    const bool wide2 = g.N2 >= g.N1;                 // orientation: the wider side gets the wide tile edge (the 4 wave columns)
    const int narrow = wide2 ? g.N1 : g.N2, wideN = wide2 ? g.N2 : g.N1;
    // tile = (narrow edge, wide edge) with the least padding: 96 / 192 x 384 (ConvNeXt-T widths), 128 / 256 x 256 (ConvNeXt-B widths)
    // (largest tile first: on equal padding the bigger accumulator tile wins - 384 x 1536 runs on 192 x 384 tiles, not 96 x 384)
    static const int cfgs[4][2] = {{192, 384}, {256, 256}, {96, 384}, {128, 256}};
    int best = -1;
    double best_waste = 1e9;
    for (int i = 0; i < 4; ++i) {
        const double w = (double)(cdiv(narrow, cfgs[i][0]) * cfgs[i][0]) * (cdiv(wideN, cfgs[i][1]) * cfgs[i][1]) / ((double)narrow * wideN);
        if (w < best_waste - 1e-9) { best_waste = w; best = i; }
    }
    static const int allow_b = getenv("MMG_TN_WIDE_B") ? atoi(getenv("MMG_TN_WIDE_B")) : 1;     // 0: the 256-wide tiles off (A/B runs)
    if (!allow_b && (best == 1 || best == 3)) {      // the 256-wide tiles off: the 384-wide ones or nothing
        best = narrow <= 96 ? 2 : 0;
        best_waste = (double)(cdiv(narrow, cfgs[best][0]) * cfgs[best][0]) * (cdiv(wideN, 384) * 384) / ((double)narrow * wideN);
    }
    if (best_waste > 1.2) return false;              // badly fitting widths stay on the 128-wide tiles of gemm_bf16.hip
    const int tn = cfgs[best][0];
    // N1 > N2 (dW1 = dh^T x of a CNBlock: [4C, C]) runs as its transpose: operands exchanged, tile flushed transposed, bias sums taken
    // from the B fragments.  Same-run A/B against the mirrored instantiations <384, 192> / <384, 96> (profiles/r02_tn_wide_swap_ab.txt):
    // equal on the stage-1/2 shapes, 2-3 % faster on 1536 x 384; MMG_TN_WIDE_MIRROR=1 brings the mirrored ones back.
    static const int mirror = getenv("MMG_TN_WIDE_MIRROR") ? atoi(getenv("MMG_TN_WIDE_MIRROR")) : 0;
    g.swapped = 0;
    if (!wide2 && !(mirror && (best == 0 || best == 2))) {
        const bf16_t* t = g.A; g.A = g.B; g.B = t;
        int x = g.N1; g.N1 = g.N2; g.N2 = x;
        x = g.lda; g.lda = g.ldb; g.ldb = x;
        g.swapped = 1;
    }
    if (g.swapped || wide2) {
        if (best == 0) launch_tw<192, 384>(g, stream);
        else if (best == 1) launch_tw<256, 256>(g, stream);
        else if (best == 2) launch_tw<96, 384>(g, stream);
        else launch_tw<128, 256>(g, stream);
    } else {
        if (tn == 96) launch_tw<384, 96>(g, stream); else launch_tw<384, 192>(g, stream);
    }
    return true;
}
