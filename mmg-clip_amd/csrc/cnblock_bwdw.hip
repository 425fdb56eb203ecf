// ConvNeXt block MLP backward with the weight gradients accumulated ON CHIP (C = 96: ConvNeXt-T stage 1).
//
// Replaces, for torchvision's CNBlock (reference: mmgclip/networks/encoder.py:53 runs `model.features`; module tree in
// notebooks/clf_convnext_tiny_experimental.ipynb cell 3: LayerNorm -> Linear(C,4C) -> GELU -> Linear(4C,C) -> layer_scale -> residual),
// the three launches of the round-1/2 backward - cnblock_mlp_bwd_kernel<96> (data path; writes g = GELU(h) and dh, 2 x [M,4C]) and
// the two gemm_tn_wide_kernel<96,384> weight-gradient GEMMs that read them back: 70 GB of HBM traffic per block at 256 x 256 x 256
// pixels, 52 GB of it the 4C-wide g / dh round trip.  Here nothing 4C-wide exists in HBM: the kernel reads dy and d (the depthwise
// output) once, writes the gradient w.r.t. d once (6 M C bytes), and leaves through fp32 atomics of the weight / bias gradients once
// per workgroup.
//
// Work split (one 8-wave workgroup per CU, persistent over 64-row tiles):
//   * the HIDDEN dimension is split over the waves: wave w owns hidden units 48w .. 48w+47 for every row.  Its slices of
//     dW2 = dy^T g ([96 x 48]) and dW1 = dh^T xhat ([48 x 96]) are 2 x 72 fp32 accumulator registers that stay put for the whole
//     launch, the matching slices of the two weight matrices (W1 gamma_ln, gamma_ls W2^T; 9 KB each) are read as ready-made MFMA
//     fragments from L2 (pre-packed by mmg_cnblock_bwdw_pack), and the rows of a tile come from LDS images that all waves share;
//   * orientation: h = xhat (W1 gamma)^T is computed as D[row][hidden] (A = rows, B = weights), so a lane holds ONE hidden unit and
//     4 consecutive rows per accumulator: bias + GELU / GELU' in place, and two such 16-row tiles ARE an MFMA operand whose k index is
//     the row - the operand of the weight-gradient products - with no LDS round trip (guide: "an accumulator tile as the next MFMA's
//     operand").  The other operand (dy^T / xhat^T, k = row) is a transposed read (ds_read_b64_tr_b16) of the same LDS row images the
//     first two products read row-wise; a row-position permutation (quads 0 and 1 of every 16 rows exchanged) plus a pitch of
//     72 rows per 16-byte column makes BOTH kinds of read conflict-free;
//   * d LN-out = dh W1 contracts over the hidden index, which sits on the lanes: dh (bf16) goes through one LDS image
//     [row tile][row quad][hidden] (8-byte pieces, conflict-free writes and transposed reads), and after a barrier every wave forms a
//     [16 rows x 48 columns] piece of the product over ALL 384 hidden units against the W1 image that lives in LDS for the whole launch;
//   * the LayerNorm backward runs on that product: gamma / beta gradients in the MFMA layout (xhat by one transposed read), the rows
//     through an fp32 LDS tile in a row-per-8-lanes layout with 16-byte global accesses.
// LayerNorm is folded: the row images hold xhat = (d - mean) rstd in bf16, gamma_ln multiplies the packed W1 of the first product and
// beta enters through the bias b1' = b1 + W1 beta; the weight gradient is un-folded at the end (dW1 = gamma (dh^T xhat) + db1 beta^T).
#include "common.h"
#include <stdlib.h>


template <int C> struct BwCfg {
    static constexpr int H4 = 4 * C;               // hidden width
    static constexpr int R = 64;                   // rows per tile
    static constexpr int KS = C / 32;              // k-steps of the C-deep products (3)
    static constexpr int CT = C / 16;              // 16-column tiles of a C-wide result (6)
    static constexpr int KG = C / 8;               // 16-byte columns of a row image (12)
    static constexpr int RP = R + 8;               // row pitch of the row images: 72 = 8 mod 16
    static constexpr int NP1 = H4 + 8;             // hidden pitch of the W1 image and of the dh image: 392 = 8 mod 16
    static constexpr int W1IMG = KG * NP1 * 16;    // bytes
    static constexpr int ROWIMG = KG * RP * 16;
    static constexpr int DHIMG = (R / 16) * 4 * NP1 * 8;
    static constexpr int OFF_W1 = 0, OFF_X = W1IMG, OFF_DY = OFF_X + ROWIMG, OFF_DH = OFF_DY + ROWIMG;
    static constexpr int OFF_LNW = OFF_DH + DHIMG, OFF_B1 = OFF_LNW + C * 4, OFF_STAT = OFF_B1 + H4 * 4;
    static constexpr int LDS = OFF_STAT + 2 * R * 2 * 4;
    // packed weight buffer (bf16 elements): [W1 LDS image][W1 gamma fragments][gamma_ls W2^T fragments]
    static constexpr long PK_W1IMG = (long)KG * NP1 * 8;
    static constexpr long PK_FRAGS = (long)(H4 / 16) * KS * 64 * 8;
    static constexpr long PK_TOTAL = PK_W1IMG + 2 * PK_FRAGS;
    static constexpr int DLP = C + 4;              // pitch (floats) of the fp32 d LN-out tile: +4 makes the LayerNorm-backward row reads conflict-free
    static_assert(R * DLP * 4 <= DHIMG, "the fp32 d LN-out tile aliases the dh image");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

struct BwArgs {
    const bf16_t* dy; const bf16_t* xd;
    const float* ln_w; const float* ln_b; float eps;
    const bf16_t* packed; const float* b1f;
    bf16_t* dd;
    float* dW1; float* db1; float* dW2raw; float* db2raw; float* ln_dw; float* ln_db;
    long M; int ntiles;
};

// position of row r (0..15) inside its 16-row group of a row image: quads 0 and 1 exchanged (see the header)
__device__ __forceinline__ int bw_pos(int r) { return r < 8 ? (r ^ 4) : r; }

// scalar byte offset of a row tile for the buffer instructions: UNSIGNED arithmetic (tile * 12 288 passes 2^31 at M = 11.2 M rows; the
// hardware reads soffset as unsigned, a signed product would be undefined behaviour), cast only at the builtin
__device__ __forceinline__ int bw_soff(unsigned tile, unsigned bytes_per_tile) { return (int)(tile * bytes_per_tile); }

__device__ __forceinline__ void bw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((address_space(3))) bf16x4 bw_lds_v4;
__device__ __forceinline__ bf16x4 bw_tr(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bw_lds_v4*)p); }
__device__ __forceinline__ bf16x8 bw_join(const bf16x4 lo, const bf16x4 hi) {
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ float bw_sum8(float v) {        // sum over the 8 lanes {8k .. 8k+7}
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    return v;
}

#ifdef BW_PROBE
// Debug builds only (tools/bwdw_probe.py): shader-clock totals per phase, summed over waves; [mode][16]
//   0 tile loop  1 P0  2 wait A  3 P1 products  4 P1 loads issue  5 P1 GELU  6 P1 weight-gradient  7 wait B  8 P2  9 wait C1+C2  10 P3  11 waves
__device__ unsigned long long g_bw_probe[2][16];
MMG_API int mmg_debug_bwdw_probe(unsigned long long* out32, int reset) {
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_bw_probe), 256) != hipSuccess) return 1;
    unsigned long long z[32] = {0};
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_bw_probe), z, 256) != hipSuccess) return 1;
    return 0;
}
#define BWP_T(var) const long long var = __builtin_readcyclecounter()
#define BWP_ADD(idx, a, b) pr[idx] += (b) - (a)
#else
#define BWP_T(var)
#define BWP_ADD(idx, a, b)
#endif

// MODE 1 ("cnblock_bwdw_w2_kernel"):  h = xhat (W1 gamma)^T + b1', g = GELU(h), dW2raw += dy^T g, db2raw += colsum(dy)
// MODE 2 ("cnblock_bwdw_dx_kernel"):  h, dG = dy (gamma_ls W2), dh = dG GELU'(h), dW1 += un-folded dh^T xhat, db1 += colsum(dh),
//                                     d LN-out = dh W1, LayerNorm backward -> dd, ln_dw, ln_db
// Two launches instead of one because the two weight-gradient slices of a wave are 2 x 72 accumulator registers: together with the
// ~100 live registers of the data path they do not fit the 256 a wave has at two waves per SIMD (hipcc spilled 36 ... 90 accumulator
// registers to scratch in every arrangement tried - AGPR-pinned accumulators included, which the compiler splits 128 / 128), and one
// wave per SIMD halves the VALU issue rate this GELU-bound kernel lives on.  The split costs one more h product (6 instead of 5
// GEMM-equivalents per block) and no extra GELU work: GELU runs in launch 1, GELU' in launch 2.
// NW = waves per workgroup: 8 (two per SIMD, 256 registers each) or 12 (three per SIMD, 168 registers: launch 1, whose register
// needs allow it - a third wave per SIMD to fill the other two's waits).  A wave owns H4 / NW hidden units.
// Lane id from the hardware, opaque to the optimiser: per-lane constants that are needed ONCE per tile (staging offsets, the row phase's row /
// part) are re-derived from it at their use instead of living in registers for the whole kernel - launch 2 runs at 256 VGPRs and hipcc
// had spilled nine of them, each reload sitting behind an `s_waitcnt vmcnt(0)` that also drained the prefetched rows and weight fragments.
__device__ __forceinline__ int bw_fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// VAR (round 4) selects the schedule of launch 1's P1 (launch 2: 0 only):
//   0  round 3: weight fragments of every step re-read from L2 (3 KiB per wave and step: with the eight waves in lockstep that is
//      24 KiB per step through the CU's 64 B/clk vector-memory path - the "load issue" segment of the phase probe), bias added at GELU time
//   1  the wave's 9 W1 fragments (36 registers) and its three bias QUADS (the first product's C operand: no bias add) stay in
//      registers for the whole launch; the bias-gradient column sums of dy only in the wave that keeps them
//   2  = 1 + software pipeline: step st's GELU block is issued BETWEEN the six h products of step st + 1 and the six weight-gradient
//      products of step st - 1 (sched_group_barrier: one MFMA per eight vector instructions), so the matrix pipe works under the
//      vector stream of the same wave instead of alternating with it
template <int C, int MODE, int NW, int VAR>
__global__ __launch_bounds__(NW * 64, NW / 4) void cnblock_bwdw_kernel(const BwArgs a) {
    typedef BwCfg<C> Cfg;
    constexpr int H4 = Cfg::H4, KS = Cfg::KS, CT = Cfg::CT, RP = Cfg::RP, NP1 = Cfg::NP1, R = Cfg::R;
    constexpr int BW_THREADS = NW * 64, HS = H4 / NW, HT = HS / 16;
    static_assert(HS % 16 == 0 && (MODE == 1 || NW == 8), "hidden slice of a wave: whole 16-unit tiles; launch 2 is laid out for 8 waves");
    // (launch 2: VAR bit 0 selects the ht-outer step order; the resident-weight schedules are launch 1's)
    constexpr int VARX = VAR;
    constexpr bool RESW = MODE == 1 && VAR >= 1, PIPE = MODE == 1 && VAR >= 2;
    constexpr bool W2 = MODE == 1, DX = MODE == 2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1img = smem + Cfg::OFF_W1;
    // launch 1 double-buffers the two row images (it has the LDS: no W1 / dh image), which leaves ONE barrier per tile; launch 2 has one set
    char* ximg = smem + Cfg::OFF_X;
    char* dyimg = smem + Cfg::OFF_DY;
    char* dhimg = smem + Cfg::OFF_DH;
    float* s_dln = reinterpret_cast<float*>(smem + Cfg::OFF_DH);           // [R][DLP] fp32 (times gamma), aliases the dh image between barriers
    float* s_lnw = reinterpret_cast<float*>(smem + Cfg::OFF_LNW);
    float* s_b1 = reinterpret_cast<float*>(smem + Cfg::OFF_B1);            // b1' = b1 + W1 beta; reused for the db1 exchange at the end
    float* s_stat = reinterpret_cast<float*>(smem + Cfg::OFF_STAT);        // [parity][R][mean, rstd]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;

    // ---- prologue: the W1 image (global layout == LDS layout; launch 2 only), LayerNorm weight, folded bias --------------------
    {
        if (DX) {
            const uint4* src = reinterpret_cast<const uint4*>(a.packed);
            for (int i = tid; i < Cfg::W1IMG / 16; i += BW_THREADS) *reinterpret_cast<uint4*>(w1img + i * 16) = src[i];
        }
        for (int i = tid; i < C; i += BW_THREADS) s_lnw[i] = a.ln_w[i];
        for (int i = tid; i < H4; i += BW_THREADS) s_b1[i] = a.b1f[i];
    }
    // Global operands go through buffer instructions: descriptor + uniform offset in scalar registers, one 32-bit lane offset - no
    // 64-bit per-lane addresses (round 3, first version: hipcc kept 18 of them, spilled to scratch, and re-loaded each behind a
    // `s_waitcnt vmcnt(0)`: 19 % of the second launch).  Byte offsets stay below 2^32 (checked by the entry point).
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, (int)(Cfg::PK_TOTAL * 2), 0x27000);
    const int w1g = (int)(Cfg::PK_W1IMG * 2) + wave * (HT * KS * 1024);               // this wave's fragments (1 KiB each)
    const int w2g = w1g + (int)(Cfg::PK_FRAGS * 2);
    // staging role of this wave: waves 0-3 bring 16 rows of d each (LayerNorm statistics, 4 lanes per row), waves 4-7 the same rows of dy
    const __amdgpu_buffer_rsrc_t rs_dd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dd, 0, (int)(unsigned)(a.M * C * 2), 0x27000);
    const bool st_dy = (wave & 4) != 0, st_on = wave < 8;          // (waves 8 .. of a 12-wave workgroup do not stage)
    const __amdgpu_buffer_rsrc_t rs_row = __builtin_amdgcn_make_buffer_rsrc((void*)(st_dy ? a.dy : a.xd), 0, (int)(unsigned)(a.M * C * 2), 0x27000);

    // ---- per-lane addresses -----------------------------------------------------------------------------------------------------
    // staging (P0): lanes 0-31 take d, lanes 32-63 take dy; 8 rows per wave, 4 lanes per row, 3 pieces of 16 bytes per lane
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);                                   // (uniform: scalar register)
    auto st_row_f = [&]() { return 16 * (wave_u & 3) + (bw_fresh_lane() & 15); };
    auto st_lds_f = [&]() { const int l = bw_fresh_lane(), r = 16 * (wave_u & 3) + (l & 15); return (((l >> 4) * RP) + 16 * (r >> 4) + bw_pos(r & 15)) * 16; };   // + ks * 4 * RP * 16
    auto st_off_f = [&]() { const int l = bw_fresh_lane(); return ((16 * (wave_u & 3) + (l & 15)) * C + 8 * (l >> 4)) * 2; };   // bytes; + tile * R * C * 2 (scalar) + 64 * ks
    // row-wise operand fragment (A of the C-deep products): row 16 rt + li, 16-byte column 4 ks + lg
    const int rd_row = (lg * RP + bw_pos(li)) * 16;                                            // + ks * 4 * RP * 16 + rt * 256
    // transposed reads of a row image: rows 4 lg + q of a 16-row tile, column piece p of a 16-column tile
    const int tr_row = ((p >> 1) * RP + bw_pos(4 * lg + q)) * 16 + (p & 1) * 8;               // + ct * 2 * RP * 16 + rt * 256
    // dh image: write 8 bytes = rows 4 lg .. +3 of hidden unit (own) ; transposed read: hidden 4 lg + q (+16), row piece p
    const int dh_wr = (lg * NP1 + HS * wave + li) * 8;                                          // + (rt * 4 * NP1 + 16 * ht) * 8
    const int dh_rd = (p * NP1 + 4 * lg + q) * 8;                                               // + rt * 4 * NP1 * 8 + (32 ks [+16]) * 8
    // W1 image, transposed: hidden 4 lg + q (+16), column piece p of column tile ct
    const int w1_rd = ((p >> 1) * NP1 + 4 * lg + q) * 16 + (p & 1) * 8;                       // + ct * 2 * NP1 * 16 + (32 ks [+16]) * 16

    f32x4 wacc[HT][CT];                        // MODE 1: dW2raw[c tile ct][hidden tile ht]; MODE 2: (dh^T xhat)[hidden tile ht][c tile ct]
#pragma unroll
    for (int i = 0; i < HT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) wacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bacc[3] = {0.f, 0.f, 0.f};          // MODE 1: [0] = column sums of dy (wave w < 6: columns 16 w ..); MODE 2: db1 of the 3 hidden tiles
    float dgacc[CT / 2] = {0.f, 0.f, 0.f}, dbacc[CT / 2] = {0.f, 0.f, 0.f};

    bw_barrier();

    u32x4_t pre[KS];
    if (st_on && (int)blockIdx.x < a.ntiles) {
        const int st_off = st_off_f();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) pre[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_row, st_off + 64 * ks, bw_soff(blockIdx.x, R * C * 2), 0);
    }

#ifdef BW_PROBE
    long long pr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    BWP_T(pr_loop0);
    bf16x8 w1f[KS], w2f[DX ? KS : 1];
    bf16x8 xa[KS][2], da[DX ? KS : 1][2];
    bf16x8 tT[CT];
    const char* timg = W2 ? dyimg : ximg;           // (re-pointed per tile in launch 1)
    auto load_w = [&](int ht) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            w1f[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, w1g + (ht * KS + ks) * 1024, 0));
            if (DX) w2f[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, w2g + (ht * KS + ks) * 1024, 0));
        }
    };
    auto load_a = [&](int rp) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int off = rd_row + ks * (4 * RP * 16) + (2 * rp + t2) * 256;
                xa[ks][t2] = *reinterpret_cast<const bf16x8*>(ximg + off);
                if (DX) da[ks][t2] = *reinterpret_cast<const bf16x8*>(dyimg + off);
            }
    };
    auto load_t = [&](int rp) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int off = tr_row + ct * (2 * RP * 16) + (2 * rp) * 256;
            tT[ct] = bw_join(bw_tr(timg + off), bw_tr(timg + off + 256));
        }
    };
    bf16x8 w1r[RESW ? HT : 1][KS];                    // VAR >= 1: this wave's W1 gamma fragments, resident
    f32x4 b1q[RESW ? HT : 1];                         //           and its bias as the first product's C operand
    if constexpr (RESW) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                w1r[ht][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, w1g + (ht * KS + ks) * 1024, 0));
            const float b = s_b1[HS * wave + 16 * ht + li];
            b1q[ht] = f32x4{b, b, b, b};
        }
    } else if ((int)blockIdx.x < a.ntiles) load_w(0);        // (the weight fragments of step 0: requested one tile ahead from here on)
    // bias gradient of the second linear = column sums of dy, taken from the transposed fragments of a row-tile pair: wave w < 6 keeps
    // columns 16 w .. (uniform branches: only the wave that keeps a column tile unpacks it)
    auto dy_colsum = [&]() {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            if (ct == wave_u) {
                float sy = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sy += bf2f((bf16_t)tT[ct][e]);
                bacc[0] += sy;
            }
    };
    auto stage_rows = [&](char* ximg_s, char* dyimg_s, int par_s) {
            char* img = st_dy ? dyimg_s : ximg_s;
            const int st_lds = st_lds_f();
            if (!st_on) {
            } else if (!st_dy) {
                // (three passes over the 12 packed registers instead of 24 unpacked floats)
                float s = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned w[4] = {pre[ks].x, pre[ks].y, pre[ks].z, pre[ks].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) s += bf2f_lo(w[e]) + bf2f_hi(w[e]);
                }
                s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
                const float mean = s * (1.0f / C);
                float qq = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned w[4] = {pre[ks].x, pre[ks].y, pre[ks].z, pre[ks].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d0 = bf2f_lo(w[e]) - mean, d1 = bf2f_hi(w[e]) - mean; qq = fmaf(d0, d0, qq); qq = fmaf(d1, d1, qq); }
                }
                qq += __shfl_xor(qq, 16, 64); qq += __shfl_xor(qq, 32, 64);
                const float rstd = rsqrtf(qq * (1.0f / C) + a.eps);
                if (DX && lg == 0) { const int st_row = st_row_f(); s_stat[(par_s * R + st_row) * 2] = mean; s_stat[(par_s * R + st_row) * 2 + 1] = rstd; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned w[4] = {pre[ks].x, pre[ks].y, pre[ks].z, pre[ks].w};
                    unsigned o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack2bf((bf2f_lo(w[e]) - mean) * rstd, (bf2f_hi(w[e]) - mean) * rstd);
                    *reinterpret_cast<uint4*>(img + st_lds + ks * (4 * RP * 16)) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<u32x4_t*>(img + st_lds + ks * (4 * RP * 16)) = pre[ks];
            }
    };
    int parity = 0;
    if constexpr (W2 && RESW) {          // the first tile's rows
        if ((int)blockIdx.x < a.ntiles) stage_rows(smem + Cfg::OFF_X, smem + Cfg::OFF_X + Cfg::ROWIMG, 0);
    }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, parity ^= 1) {
        BWP_T(pr_p0);
        if (W2) {                 // this tile's image pair
            ximg = smem + Cfg::OFF_X + parity * (2 * Cfg::ROWIMG);
            dyimg = ximg + Cfg::ROWIMG;
            timg = dyimg;
        }
        // ================= P0: rows -> LDS images (xhat, dy), row statistics =======================================================
        // (VAR >= 1, launch 1: this tile's images were written in the MIDDLE of the previous tile's P1 - see there; the first tile's before the loop)
        if (!(W2 && RESW)) stage_rows(ximg, dyimg, parity);
        BWP_T(pr_a0);
        bw_barrier();                                                                                    // A
        BWP_T(pr_a1);
        BWP_ADD(1, pr_p0, pr_a0); BWP_ADD(2, pr_a0, pr_a1);
        // next tile's rows, consumed at its P0.  Launch 1 requests them now (it stages them in the middle of this tile's P1).  Launch 2 requests them
        // AFTER its P1 (round 4, ISA read): its step loop waits for weight fragments requested during the loop, the vector-memory counter retires in
        // order, and hipcc's `s_waitcnt vmcnt(0)` in front of the second step's products therefore also waited for these three HBM loads - one
        // exposed memory latency per tile; behind the loop they have P2 + P3 to land, and P1 has 12 registers more.
        auto request_rows = [&]() {
            if (st_on && tile + (int)gridDim.x < a.ntiles) {
                const int st_off_n = st_off_f();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    pre[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_row, st_off_n + 64 * ks, bw_soff((unsigned)tile + gridDim.x, R * C * 2), 0);
            }
        };
        if (!DX) request_rows();

        // ================= P1: this wave's 48 hidden units over the 64 rows ==========================================================
        // Six steps (row-tile pair rp, hidden tile ht), rp outer.  Every operand is requested well before its use and lives in ONE
        // register set that is re-loaded right after its last use (hipcc, left alone, emitted `ds_read; s_waitcnt lgkmcnt(0); mfma` per
        // MFMA on a single fragment register: 58 % of the wave time parked at waits, PMC round 3):
        //   row fragments (xhat, dy rows of this rp)   once per rp, loaded after the last product of the previous rp
        //   transposed fragments (k = rows)            once per rp, loaded after the last weight-gradient product of the previous rp
        //   weight fragments of the next step          requested from L2 right after this step's products, landing under the GELU block
        if constexpr (PIPE) {
            // ---- launch 1, VAR 2: one-step software pipeline.  Program order of step st: the six weight-gradient products of step
            // st - 1 (operand: the g fragment kept from it), the six h products of step st + 1 (C operand = the bias quad), then step
            // st's GELU block; sched_group_barrier spreads the twelve MFMAs one per seven vector instructions, so that the matrix pipe
            // runs under this wave's own vector stream.  Row fragments of the next rp are requested at the top of the step that issues
            // its first h products; the transposed fragments of rp 1 after the last weight-gradient product of rp 0 has been issued.
            load_a(0); load_t(0);
            f32x4 hb[2][2];
            bf16x8 ofp = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            auto h_prod = [&](int ht_n, f32x4 (&h)[2]) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2)
                        h[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[ks][t2], w1r[RESW ? ht_n : 0][ks], ks == 0 ? b1q[RESW ? ht_n : 0] : h[t2], 0, 0, 0);
            };
            h_prod(0, hb[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 2 * HT; ++st) {
                const int rp = st / HT, ht = st % HT;
                if constexpr (W2 && RESW) {
                    // the NEXT tile's rows (requested after barrier A, 3 steps ago) go into the other image pair here, in the middle of this
                    // tile's P1: the LayerNorm statistics of waves 0-3 run beside their SIMD partners' GELU blocks instead of in front of a
                    // barrier every wave waits at (phase probe, round 3: P0 + wait A = 15 % of the launch).  Legal without another barrier:
                    // that pair was last read in the previous tile's P1, which every wave left before it passed this tile's barrier A.
                    if (st == HT && tile + (int)gridDim.x < a.ntiles) {
                        char* nx = smem + Cfg::OFF_X + (parity ^ 1) * (2 * Cfg::ROWIMG);
                        stage_rows(nx, nx + Cfg::ROWIMG, parity ^ 1);
                    }
                }
                if (st == 0 || st == HT + 1) dy_colsum();                       // (tT holds rp 0 / rp 1 from here: rp 1 is requested at the end of step HT)
                if (ht == HT - 1 && rp == 0) load_a(1);                         // xa of rp 0: last read by the h products issued in the previous step
                __builtin_amdgcn_sched_barrier(0);
                if (st > 0) {                                                   // weight gradient of step st - 1
                    const int htp = (st - 1) % HT;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) wacc[htp][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tT[ct], ofp, wacc[htp][ct], 0, 0, 0);
                }
                if (st + 1 < 2 * HT) h_prod((st + 1) % HT, hb[(st + 1) & 1]);
                unsigned op[4];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_bf16(hb[st & 1][t2][e]);
                    op[2 * t2] = pack2bf(v[0], v[1]); op[2 * t2 + 1] = pack2bf(v[2], v[3]);
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);          // seven VALU
                }
                __builtin_amdgcn_sched_barrier(0);
                ofp = __builtin_bit_cast(bf16x8, (u32x4_t{op[0], op[1], op[2], op[3]}));
                if (st == HT) load_t(1);                                        // rp 0's last weight-gradient product was issued above
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wacc[HT - 1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tT[ct], ofp, wacc[HT - 1][ct], 0, 0, 0);
        } else
        {
            load_a(0); load_t(0);
            float dbsum[3] = {0.f, 0.f, 0.f};
            // step order: launch 1 (rp outer) keeps the row fragments of an rp for its three steps; launch 2 re-reads its row fragments every step
            // anyway (no registers to keep them), so it runs ht OUTER (MMG_BWDW_HTO, round 4): the two weight fragment sets of a hidden tile are
            // fetched from L2 once per tile instead of once per row pair - half of the "load issue" segment of the phase probe (48 KiB per step and
            // CU through the 64 B/clk vector-memory path) - at the price of the transposed xhat fragments being re-read from LDS every step
            constexpr bool HTO = DX && (VARX & 1);
#pragma unroll
            for (int st = 0; st < 2 * HT; ++st) {
                const int rp = HTO ? st % 2 : st / HT, ht = HTO ? st / 2 : st % HT;
                if constexpr (W2 && RESW) {
                    // the NEXT tile's rows (requested after barrier A, 3 steps ago) go into the other image pair here, in the middle of this
                    // tile's P1: the LayerNorm statistics of waves 0-3 run beside their SIMD partners' GELU blocks instead of in front of a
                    // barrier every wave waits at (phase probe, round 3: P0 + wait A = 15 % of the launch).  Legal without another barrier:
                    // that pair was last read in the previous tile's P1, which every wave left before it passed this tile's barrier A.
                    if (st == HT && tile + (int)gridDim.x < a.ntiles) {
                        char* nx = smem + Cfg::OFF_X + (parity ^ 1) * (2 * Cfg::ROWIMG);
                        stage_rows(nx, nx + Cfg::ROWIMG, parity ^ 1);
                    }
                }
                const float bias = RESW ? 0.f : s_b1[HS * wave + 16 * ht + li];     // (VAR >= 1: the bias is the products' C operand)
                f32x4 hacc[2], gacc[2];
                BWP_T(pr_s0);
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) { hacc[t2] = RESW ? b1q[RESW ? ht : 0] : f32x4{0.f, 0.f, 0.f, 0.f}; gacc[t2] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        hacc[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[ks][t2], RESW ? w1r[RESW ? ht : 0][ks] : w1f[ks], hacc[t2], 0, 0, 0);
                        if (DX) gacc[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[ks][t2], w2f[ks], gacc[t2], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                BWP_T(pr_s1);
                if (st + 1 == 2 * HT) {
                    if (!RESW && tile + (int)gridDim.x < a.ntiles) load_w(0);      // next tile's first step: in flight over P2 / P3 / P0
                } else {
                    if (!RESW && (!HTO || rp == 1)) load_w(HTO ? ht + 1 : (st + 1) % HT);
                    if (W2 && ht == HT - 1) load_a(rp + 1);      // launch 1 keeps the row fragments of an rp for its three steps
                }
                __builtin_amdgcn_sched_barrier(0);
                BWP_T(pr_s2);
                // GELU (launch 1) / GELU' (launch 2) in place: lane = hidden unit 48 w + 16 ht + li, rows 16 (2 rp + t2) + 4 lg + e
                unsigned op[4];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (W2) v[e] = gelu_bf16(RESW ? hacc[t2][e] : hacc[t2][e] + bias);
                        else { v[e] = gacc[t2][e] * gelu_bf16_grad_poly(hacc[t2][e] + bias); dbsum[ht] += v[e]; }
                    }
                    op[2 * t2] = pack2bf(v[0], v[1]); op[2 * t2 + 1] = pack2bf(v[2], v[3]);
                    if (DX) *reinterpret_cast<uint2*>(dhimg + dh_wr + ((2 * rp + t2) * 4 * NP1 + 16 * ht) * 8) = make_uint2(op[2 * t2], op[2 * t2 + 1]);
                }
                const bf16x8 of = __builtin_bit_cast(bf16x8, (u32x4_t{op[0], op[1], op[2], op[3]}));       // g (launch 1) / dh (launch 2)
                __builtin_amdgcn_sched_barrier(0);
                // launch 2 (12 row fragments = 48 registers): re-requested for every step, AFTER the GELU block whose temporaries they
                // would not fit beside, in flight under the six weight-gradient products
                BWP_T(pr_s3);
                if (DX && st + 1 < 2 * HT) load_a(HTO ? (st + 1) % 2 : (st + 1) / HT);
                // weight gradient: k = the 32 rows of this pair of row tiles (slot 8 lg + j: j < 4 row 4 lg + j, else 16 + 4 lg + j - 4)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (W2) wacc[ht][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tT[ct], of, wacc[ht][ct], 0, 0, 0);     // D[c][hidden]
                    else wacc[ht][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of, tT[ct], wacc[ht][ct], 0, 0, 0);        // D[hidden][c]
                }
                if (W2 && ht == 0) {          // bias gradient of the second linear: column sums of dy; wave w < 6 keeps columns 16 w ..
                    if constexpr (RESW) dy_colsum();
                    else {
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            float sy = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) sy += bf2f((bf16_t)tT[ct][e]);
                            bacc[0] += (ct == wave) ? sy : 0.f;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (HTO ? (st + 1 < 2 * HT) : (ht == HT - 1 && rp == 0)) load_t(HTO ? (st + 1) % 2 : 1);
                BWP_T(pr_s4);
                BWP_ADD(3, pr_s0, pr_s1); BWP_ADD(4, pr_s1, pr_s2); BWP_ADD(5, pr_s2, pr_s3); BWP_ADD(6, pr_s3, pr_s4);
            }
            if (DX) {
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) bacc[ht] += dbsum[ht];
            }
        }
        if (DX) request_rows();
        BWP_T(pr_b0);
        if (DX) bw_barrier();                                                                            // B: the dh image is complete (launch 1: nothing to wait for - the next tile stages into the other image pair)
        BWP_T(pr_b1);
        BWP_ADD(7, pr_b0, pr_b1);
        if (DX) {
            // ================= P2: d LN-out = dh W1 for (row tile rt, 3 column tiles), gamma / beta gradients ========================
            const int rt = wave >> 1, ch = wave & 1;
            f32x4 lacc[CT / 2];
#pragma unroll
            for (int c3 = 0; c3 < CT / 2; ++c3) lacc[c3] = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                const char* pa0 = dhimg + dh_rd + rt * (4 * NP1 * 8);
                const char* pb0 = w1img + w1_rd + (3 * ch) * (2 * NP1 * 16);
                bf16x8 af = bw_join(bw_tr(pa0), bw_tr(pa0 + 16 * 8));
                bf16x8 bf[CT / 2];
#pragma unroll
                for (int c3 = 0; c3 < CT / 2; ++c3) bf[c3] = bw_join(bw_tr(pb0 + c3 * (2 * NP1 * 16)), bw_tr(pb0 + c3 * (2 * NP1 * 16) + 16 * 16));
#pragma unroll
                for (int ks = 0; ks < H4 / 32; ++ks) {
                    const bf16x8 ac = af;
                    bf16x8 bc[CT / 2];
#pragma unroll
                    for (int c3 = 0; c3 < CT / 2; ++c3) bc[c3] = bf[c3];
                    if (ks + 1 < H4 / 32) {               // next k-step's fragments are in flight under this step's products
                        const char* pa = pa0 + (ks + 1) * (32 * 8);
                        af = bw_join(bw_tr(pa), bw_tr(pa + 16 * 8));
#pragma unroll
                        for (int c3 = 0; c3 < CT / 2; ++c3) {
                            const char* pb = pb0 + c3 * (2 * NP1 * 16) + (ks + 1) * (32 * 16);
                            bf[c3] = bw_join(bw_tr(pb), bw_tr(pb + 16 * 16));
                        }
                    }
#pragma unroll
                    for (int c3 = 0; c3 < CT / 2; ++c3) lacc[c3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac, bc[c3], lacc[c3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int c3 = 0; c3 < CT / 2; ++c3) {
                const bf16x4 xh = bw_tr(ximg + tr_row + (3 * ch + c3) * (2 * RP * 16) + rt * 256);          // xhat[16 rt + 4 lg + e][16 ct + li]
                float sg = 0.f, sb = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { sb += lacc[c3][e]; sg = fmaf(lacc[c3][e], bf2f((bf16_t)xh[e]), sg); }
                dgacc[c3] += sg; dbacc[c3] += sb;
            }
            BWP_T(pr_c0);
            // P3's operands that live in the row images: this thread's 16 xhat values (row tid >> 3, columns 16 (tid & 7) ..), read now -
            // the next tile's P0 may overwrite the images as soon as the last wave has passed C2
            const int p3_tid = 64 * wave_u + bw_fresh_lane();
            const int p3_row = p3_tid >> 3, p3_part = (p3_tid & 7) < CT ? (p3_tid & 7) : 0;
            const u32x4_t xq0 = *reinterpret_cast<const u32x4_t*>(ximg + ((2 * p3_part) * RP + 16 * (p3_row >> 4) + bw_pos(p3_row & 15)) * 16);
            const u32x4_t xq1 = *reinterpret_cast<const u32x4_t*>(ximg + ((2 * p3_part + 1) * RP + 16 * (p3_row >> 4) + bw_pos(p3_row & 15)) * 16);
            bw_barrier();                                                                                // C1: dh image has been read
            BWP_T(pr_c1);
#pragma unroll
            for (int c3 = 0; c3 < CT / 2; ++c3) {
                const float gm = s_lnw[16 * (3 * ch + c3) + li];          // d xhat = d LN-out * gamma: folded here, the row phase reads no gamma
#pragma unroll
                for (int e = 0; e < 4; ++e) s_dln[(16 * rt + 4 * lg + e) * Cfg::DLP + 16 * (3 * ch + c3) + li] = lacc[c3][e] * gm;
            }
            BWP_T(pr_c2);
            bw_barrier();                                                                                // C2
            BWP_T(pr_c3);
            BWP_ADD(8, pr_b1, pr_c0); BWP_ADD(9, pr_c0, pr_c1); BWP_ADD(9, pr_c2, pr_c3);

            // ================= P3: LayerNorm backward, one row per 8 lanes (6 of them active, 16 columns each) =======================
            const int row = p3_row, part = p3_tid & 7;
            const bool act = part < CT;
            const int c0 = act ? 16 * part : 0;
            const float rstd = s_stat[(parity * R + row) * 2 + 1];
            float x[16], gv[16];
            float s1 = 0.f, s2 = 0.f;
            {
                const unsigned w[8] = {xq0.x, xq0.y, xq0.z, xq0.w, xq1.x, xq1.y, xq1.z, xq1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[2 * e] = bf2f_lo(w[e]); x[2 * e + 1] = bf2f_hi(w[e]); }
            }
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const f32x4 dl = *reinterpret_cast<const f32x4*>(s_dln + row * Cfg::DLP + c0 + 4 * v4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gv[4 * v4 + e] = dl[e]; s1 += dl[e]; s2 = fmaf(dl[e], x[4 * v4 + e], s2); }
            }
            if (!act) { s1 = 0.f; s2 = 0.f; }
            s1 = bw_sum8(s1); s2 = bw_sum8(s2);
            if (act) {
                const float m1 = s1 * (1.0f / C), m2 = s2 * (1.0f / C);
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8) {
                    unsigned o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[e] = pack2bf(rstd * (fmaf(-x[8 * h8 + 2 * e], m2, gv[8 * h8 + 2 * e]) - m1), rstd * (fmaf(-x[8 * h8 + 2 * e + 1], m2, gv[8 * h8 + 2 * e + 1]) - m1));
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{o[0], o[1], o[2], o[3]}, rs_dd, (row * C + c0 + 8 * h8) * 2, bw_soff((unsigned)tile, R * C * 2), 0);
                }
            }
            // (no barrier: the next P0 writes the row images and the other parity of s_stat; s_dln is rewritten as the dh image only
            //  after the next tile's barrier A, which every wave reaches after its P3)
            BWP_T(pr_p3);
            BWP_ADD(10, pr_c3, pr_p3);
        }
    }
#ifdef BW_PROBE
    {
        BWP_T(pr_loop1);
        pr[0] = pr_loop1 - pr_loop0; pr[11] = 1;
        if (lane == 0)
            for (int i = 0; i < 12; ++i) atomicAdd(&g_bw_probe[MODE - 1][i], (unsigned long long)pr[i]);
    }
#endif

    // ================= epilogue: the accumulators leave as fp32 atomics =================================================================
    bw_barrier();
    if (W2) {
        if (wave < CT) {
            float v = bacc[0];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lg == 0) atomicAdd(a.db2raw + 16 * wave + li, v);
        }
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e)       // dW2raw[c = 16 ct + 4 lg + e][hidden 48 w + 16 ht + li]
                    atomicAdd(a.dW2raw + (long)(16 * ct + 4 * lg + e) * H4 + HS * wave + 16 * ht + li, wacc[ht][ct][e]);
    } else {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            float v = bacc[ht];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lg == 0) { s_b1[HS * wave + 16 * ht + li] = v; atomicAdd(a.db1 + HS * wave + 16 * ht + li, v); }
        }
        const int ch = wave & 1;
#pragma unroll
        for (int c3 = 0; c3 < CT / 2; ++c3) {
            float g = dgacc[c3], b = dbacc[c3];
            g += __shfl_xor(g, 16, 64); g += __shfl_xor(g, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (lg == 0) { atomicAdd(a.ln_dw + 16 * (3 * ch + c3) + li, g); atomicAdd(a.ln_db + 16 * (3 * ch + c3) + li, b); }
        }
        bw_barrier();
#pragma unroll
        for (int ht = 0; ht < HT; ++ht)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 16 * ct + li;
                const float gm = s_lnw[c], bt = a.ln_b[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // dW1[hidden 48 w + 16 ht + 4 lg + e][c] = gamma[c] (dh^T xhat) + beta[c] db1[hidden]   (un-folding of the LayerNorm affine)
                    const int n = HS * wave + 16 * ht + 4 * lg + e;
                    atomicAdd(a.dW1 + (long)n * C + c, fmaf(gm, wacc[ht][ct][e], bt * s_b1[n]));
                }
            }
    }
}

// ---- weight packing ------------------------------------------------------------------------------------------------------------------
struct BwPack { const float* w1; const float* w2; const float* ln_w; const float* ln_b; const float* ls; const float* b1; bf16_t* out; float* b1f; int C; };

template <int C>
__global__ __launch_bounds__(256) void cnblock_bwdw_pack_kernel(const BwPack a) {
    typedef BwCfg<C> Cfg;
    constexpr int H4 = Cfg::H4, NP1 = Cfg::NP1, KS = Cfg::KS;
    const long total = Cfg::PK_TOTAL;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total + H4; e += (long)gridDim.x * blockDim.x) {
        if (e >= total) {                         // b1' = b1 + W1 beta
            const int n = (int)(e - total);
            float s = a.b1[n];
            for (int c = 0; c < C; ++c) s = fmaf(a.w1[(long)n * C + c], a.ln_b[c], s);
            a.b1f[n] = s;
            continue;
        }
        float v;
        if (e < Cfg::PK_W1IMG) {                  // [16-byte column kg][hidden, pitch NP1][8]: plain W1 (the d LN-out product)
            const int kg = (int)(e / (NP1 * 8)), n = (int)((e / 8) % NP1), j = (int)(e % 8);
            v = n < H4 ? a.w1[(long)n * C + 8 * kg + j] : 0.f;
        } else {                                  // fragments: [which][16-unit hidden tile][ks][lane][8]: B[k = c = 32 ks + 8 lg + j][hidden 48 w + 16 ht + li]
            long f = e - Cfg::PK_W1IMG;
            const int which = (int)(f / Cfg::PK_FRAGS);
            f -= (long)which * Cfg::PK_FRAGS;
            const int j = (int)(f % 8), lane = (int)((f / 8) % 64), ks = (int)((f / 512) % KS), tile16 = (int)(f / (512 * KS));
            const int n = 16 * tile16 + (lane & 15), c = 32 * ks + 8 * (lane >> 4) + j;    // (wave w of an NW-wave launch owns tiles w H4/(16 NW) ..)
            v = which == 0 ? a.w1[(long)n * C + c] * a.ln_w[c]                 // W1 gamma_ln  (LayerNorm affine folded)
                           : a.w2[(long)c * H4 + n] * a.ls[c];                 // gamma_ls W2^T (layer scale folded: dG = dy (gamma W2))
        }
        a.out[e] = f2bf(v);
    }
}

#define BW_DEFAULT_VAR 2
#define BW_DEFAULT_HTO 1
static int bw_cu_count() { return mmg_cu_count_cached(); }

MMG_API int mmg_cnblock_bwdw_supported(int C) { return C == 96 ? 1 : 0; }
MMG_API long long mmg_cnblock_bwdw_packed_elems(int C) { return C == 96 ? (long long)BwCfg<96>::PK_TOTAL : 0; }

MMG_API int mmg_cnblock_bwdw_pack(const float* w1, const float* w2, const float* ln_w, const float* ln_b, const float* layer_scale,
                                  const float* b1, void* packed, float* b1f, int C, hipStream_t stream) {
    MMG_CHECK_ARG(C == 96, "mmg_cnblock_bwdw_pack: C=%d is not supported (96)", C);
    MMG_CHECK_ARG(w1 && w2 && ln_w && ln_b && layer_scale && b1 && packed && b1f, "mmg_cnblock_bwdw_pack: null pointer");
    BwPack a{w1, w2, ln_w, ln_b, layer_scale, b1, (bf16_t*)packed, b1f, C};
    hipLaunchKernelGGL(cnblock_bwdw_pack_kernel<96>, dim3(128), dim3(256), 0, stream, a);
    MMG_LAUNCH_CHECK("mmg_cnblock_bwdw_pack");
    return 0;
}

MMG_API int mmg_cnblock_bwdw(const void* dy, const void* xd, const float* ln_w, const float* ln_b, float eps, const void* packed,
                             const float* b1f, void* dd, float* dW1, float* db1, float* dW2raw, float* db2raw, float* ln_dw,
                             float* ln_db, long long M, int C, hipStream_t stream) {
    MMG_CHECK_ARG(C == 96, "mmg_cnblock_bwdw: C=%d is not supported (96)", C);
    MMG_CHECK_ARG(M > 0 && M % BwCfg<96>::R == 0, "mmg_cnblock_bwdw: M=%lld must be a positive multiple of %d", M, BwCfg<96>::R);
    MMG_CHECK_ARG(M * 96 * 2 < (1LL << 32), "mmg_cnblock_bwdw: M=%lld rows exceed the 4 GiB buffer range of one launch", M);
    MMG_CHECK_ARG(dy && xd && ln_w && ln_b && packed && b1f && dd && dW1 && db1 && dW2raw && db2raw && ln_dw && ln_db,
                  "mmg_cnblock_bwdw: null pointer");
    BwArgs a;
    a.dy = (const bf16_t*)dy; a.xd = (const bf16_t*)xd; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = eps;
    a.packed = (const bf16_t*)packed; a.b1f = b1f; a.dd = (bf16_t*)dd;
    a.dW1 = dW1; a.db1 = db1; a.dW2raw = dW2raw; a.db2raw = db2raw; a.ln_dw = ln_dw; a.ln_db = ln_db;
    a.M = M; a.ntiles = (int)(M / BwCfg<96>::R);
    const int cap = bw_cu_count();
    const int grid = a.ntiles < cap ? a.ntiles : cap;
    // launch 1 with 12 waves (three per SIMD, 168 registers): same-box A/B at 16.8 M rows, two runs each: 13.10 / 13.15 ms against
    // 12.94 / 13.09 with 8 waves - the launch is bound by VALU issue (GELU), not by latency, so the third wave buys nothing.  Off.
    static const int w12 = getenv("MMG_BWDW_W12") ? atoi(getenv("MMG_BWDW_W12")) : 0;
    // schedule of launch 1 (see the kernel's VAR comment); MMG_BWDW_VAR = 0 / 1 / 2 for same-process A/B runs (tools/bwdw_bench.py)
    const char* ev = getenv("MMG_BWDW_VAR");
    const int var = ev ? atoi(ev) : BW_DEFAULT_VAR;
    typedef BwCfg<96> Cfg;
    MMG_NOTE_KERNEL("cnblock_bwdw_kernel<96, *>");
    if (w12) {
        mmg_allow_lds(cnblock_bwdw_kernel<96, 1, 12, 0>, Cfg::LDS);
        hipLaunchKernelGGL((cnblock_bwdw_kernel<96, 1, 12, 0>), dim3(grid), dim3(768), Cfg::LDS, stream, a);
    } else if (var == 2) {
        mmg_allow_lds(cnblock_bwdw_kernel<96, 1, 8, 2>, Cfg::LDS);
        hipLaunchKernelGGL((cnblock_bwdw_kernel<96, 1, 8, 2>), dim3(grid), dim3(512), Cfg::LDS, stream, a);
    } else if (var == 1) {
        mmg_allow_lds(cnblock_bwdw_kernel<96, 1, 8, 1>, Cfg::LDS);
        hipLaunchKernelGGL((cnblock_bwdw_kernel<96, 1, 8, 1>), dim3(grid), dim3(512), Cfg::LDS, stream, a);
    } else {
        mmg_allow_lds(cnblock_bwdw_kernel<96, 1, 8, 0>, Cfg::LDS);
        hipLaunchKernelGGL((cnblock_bwdw_kernel<96, 1, 8, 0>), dim3(grid), dim3(512), Cfg::LDS, stream, a);
    }
    const char* eh = getenv("MMG_BWDW_HTO");                 // (read per call: same-process A/B runs, tools/bwdw_bench.py)
    const int hto = eh ? atoi(eh) : BW_DEFAULT_HTO;
    if (hto) {
        mmg_allow_lds(cnblock_bwdw_kernel<96, 2, 8, 1>, Cfg::LDS);
        hipLaunchKernelGGL((cnblock_bwdw_kernel<96, 2, 8, 1>), dim3(grid), dim3(512), Cfg::LDS, stream, a);
    } else {
        mmg_allow_lds(cnblock_bwdw_kernel<96, 2, 8, 0>, Cfg::LDS);
        hipLaunchKernelGGL((cnblock_bwdw_kernel<96, 2, 8, 0>), dim3(grid), dim3(512), Cfg::LDS, stream, a);
    }
    MMG_LAUNCH_CHECK("mmg_cnblock_bwdw");
    return 0;
}
