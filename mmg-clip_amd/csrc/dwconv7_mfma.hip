// Depthwise 7x7 convolution on the matrix cores (gfx950): forward and data gradient of torchvision CNBlock's
// Conv2d(dim, dim, 7, padding=3, groups=dim) - reference: mmgclip/networks/encoder.py:53 runs the ConvNeXt `features`
// (module tree in notebooks/clf_convnext_tiny_experimental.ipynb cell 3).  Round 4; the direct fp32-VALU kernels of dwconv7.hip
// stay (weight gradient, fallback, A/B: MMG_DWCONV_MFMA=0).
//
// A depthwise convolution has no channel reduction, so it is not a GEMM; but for ONE channel and ONE kernel row ky the 1-D
// convolution along x of 16 output columns is a product with a banded Toeplitz matrix:
//     out[y][x0+n] += sum_k in[y+ky-3][x0-3+k] * T_ky[k][n],    T_ky[k][n] = w[ky][k-n] if 0 <= k-n <= 6 else 0,   k < 32
// = one v_mfma_f32_16x16x32_bf16 per (channel, ky) and 16 x 16 output pixels (22 of the 32 k carry data; taps rounded to bf16).
// At 7 MFMAs per 256 outputs the matrix pipes need 0.11 cycles per output and CU against the 0.35 the VALU kernels take.
//
// What decides the speed is the LAYOUT CHANGE: the MFMA contracts over x, NHWC memory has the channel fastest.  Rounds 1-2 had this
// formulation at parity with the VALU kernel because they scattered the halo tile into channel planes with 2- / 4-byte LDS stores and
// wrote 8-byte pieces to HBM (profiles: MFMAs 30 us of 247 us).  Here both transposes are done by the LDS itself:
//   in :  HBM --16-byte loads into registers, two items ahead--> ds_write_b128 --> NHWC halo tile [22 x 22 pixels][32 ch]
//         --ds_read_b64_tr_b16 (4 pixels x 16 channels: lane = channel, 4 consecutive x)--> ds_write_b64 --> channel planes
//         [32][22 rows][48 x] (96-byte rows: conflict-free ds_read_b128 of the data operand: lane = row, 8 consecutive x)
//   out:  swapped MFMA (lane = row, 4 consecutive x_out) --ds_write_b64--> planes [32][16][16] --ds_read_b64_tr_b16 (4 channels x
//         16 x: lane = pixel, 4 consecutive channels) x 2--> ds_write_b128 --> NHWC tile --ds_read_b128--> 16-byte global stores
//         (+ the residual-gradient add of the data-gradient call)
// One 16-wave (or 8-wave) workgroup per CU, persistent over the (image, tile) items of ONE 32-channel slab; a wave owns 2 (4) channels and
// keeps their 14 (28) Toeplitz fragments (56 / 112 registers) for its whole life.  The halo tile of item k + 2 is in flight in registers
// while item k is computed (see "Memory pipeline" below).  Pitches (tools: bank model of MI355X_MICROARCH.md, LDS section): plane rows 96 B (b128 reads conflict-free under
// the real 16-lane grouping; 80 B is 2-way), planes 2128 B apart (b64 writes 2-way), output planes 40-byte rows / 656 B apart
// (writes conflict-free, transposed reads 2-way), NHWC output pixels 80 B apart (b128 writes conflict-free).
#include "common.h"
#include <stdlib.h>

#define DM_T 16                              // output tile edge
#define DM_H 22                              // halo tile edge
#define DM_CB 32                             // channels per slab (64 contiguous bytes per pixel)
#define DM_NPIX (DM_H * DM_H)                // 484
#define DM_CHUNKS 2048                       // 16-byte chunks a workgroup moves per halo tile (4 per thread; 1936 carry pixels)
#define DM_IN_BYTES (DM_CHUNKS * 16)         // the NHWC halo tile (the transposed reads of a row's last column group run 2 pixels over: inside)
#define DM_ROWB 96                           // bytes per plane row (48 bf16: x_in 0..21 data, 22..31 read by the MFMA against zero taps)
#define DM_PLANE (DM_H * DM_ROWB + 16)       // 2128
#define DM_PLANAR (DM_CB * DM_PLANE)         // 68 096
#define DM_OROWB 40                          // a channel's 16 x 16 outputs go back into ITS OWN (consumed) input plane: 40-byte rows
#define DM_OPIX 80                           // bytes per pixel of the NHWC output tile
#define DM_OUT_BYTES (DM_T * DM_T * DM_OPIX)
#define DM_OFF_IN 0
#define DM_OFF_PLANAR DM_IN_BYTES
#define DM_OFF_OUT (DM_OFF_PLANAR + DM_PLANAR)
#define DM_LDS (DM_OFF_OUT + DM_OUT_BYTES)
static_assert(DM_LDS <= 160 * 1024, "LDS budget");
static_assert(DM_PLANE % 16 == 0 && DM_T * DM_OROWB <= DM_PLANE, "alignment / the output rows fit the plane they replace");

struct DwM {
    const bf16_t* x; const float* w; const float* bias; const bf16_t* add; bf16_t* y;
    int n, H, W, C, tiles_w, tiles_h, nt;
    unsigned m_img, m_tw;   // floor(2^32 / (tiles_w * tiles_h)), floor(2^32 / tiles_w)
    int dbg;          // (unused since the ablation mask became a build constant: -DDWM_DBG=mask, tools/build_ab_lib.sh dwm<mask> dwconv7_mfma -DDWM_DBG=<mask>:
                      //  1 skip B, 2 skip D, 4 skip E, 8 skip F, 16 skip the global loads; results are then wrong)
};

// Workgroup barrier of the item loop: LDS traffic only.  __syncthreads() is fence + barrier and hipcc drains the vector-memory counter for it
// (s_waitcnt vmcnt(0)): that waited, at EVERY barrier, for the halo tiles requested two items ahead and for the previous item's stores - the
// phases of the first two versions ran strictly one after the other (ablation, tools/dwm_ablate.py: 52 us of 165 were "deposit + barriers").
__device__ __forceinline__ void dm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((address_space(3))) bf16x4 dm_lds_v4;
__device__ __forceinline__ bf16x4 dm_tr(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((dm_lds_v4*)p); }

// Memory pipeline (first version of this round: LDS-DMA one tile ahead, 9 500 cycles per tile against ~3 500 of HBM time - one 31 KB tile in
// flight per CU does not cover the latency under load).  Now the halo tile of item k + 2 is requested into REGISTERS (4 x 16 B per thread, two
// named sets by item parity) right after item k's set has been written to the LDS tile, so two tiles (62 KB per CU) are in flight over two whole
// item times; plain loads, so hipcc's own counted waits apply.  Out-of-image pixels are zeroed in registers; addresses are clamped into the image.
// NW = waves per workgroup: 8 (4 channels each, 112 registers of Toeplitz fragments, two waves per SIMD) or 16 (2 channels each, 56 registers,
// four waves per SIMD: the phases between two barriers are short dependent chains - LDS read -> MFMA chain -> LDS write - and with two waves
// per SIMD their latencies lie open).
// ADD (residual-gradient operand) and the ablation mask are COMPILE-TIME: as run-time branches they left hipcc with registers that "may be" the
// destination of a pending load on one of two merging paths, and it put `s_waitcnt vmcnt(0)` in front of the output stores of every item - of the
// forward call too, which has no such operand - draining the two-items-ahead halo prefetch (round 4, ISA read).
#ifndef DWM_DBG
#define DWM_DBG 0
#endif
template <bool FLIP, int NW, bool ADD>
__global__ __launch_bounds__(NW * 64, NW / 4) void dwconv7_mfma_kernel(const DwM p) {
    constexpr int DM_THREADS = NW * 64, CPW = DM_CB / NW, NIN = DM_CHUNKS / DM_THREADS, NOUT = 1024 / DM_THREADS;
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    // ---- which slab, which stream of items.  Workgroups b and b + 8 share an XCD (observed dispatch; speed only).  The 32 workgroups of an
    // XCD group are dealt over the slabs (slab = j % slabs): the slabs of one pixel - 64-byte pieces of the same 128-byte lines - meet in one
    // L2; and with n >= 8 every IMAGE belongs to one XCD group (img % 8), its tiles walked in order by that group's streams, so the halo
    // columns / rows that neighbouring tiles share are L2 hits instead of second HBM reads (16 x 16 tiles read 1.89 x their pixels).
    const int slabs = p.C / DM_CB;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int slab = j % slabs, u = j / slabs;
    const int ns = (per_xcd - slab + slabs - 1) / slabs;                 // streams of this slab per XCD group
    const int c0 = slab * DM_CB;
    const int per_img = p.tiles_w * p.tiles_h;
    const bool by_image = p.n >= 8;
    const int my_imgs = by_image ? (p.n - xcd + 7) / 8 : p.n;
    const int items = by_image ? my_imgs * per_img : p.n * per_img;      // length of the sequence this workgroup's streams share
    const int first = by_image ? u : u * 8 + xcd, step = by_image ? ns : 8 * ns;
    // (uniform arithmetic; the two divisions by launch constants as multiply-high + one correction instead of ~40 instructions each)
    auto udiv = [](unsigned n, unsigned d, unsigned m) { unsigned q = __umulhi(n, m); if (n - q * d >= d) ++q; return (int)q; };   // m = floor(2^32 / d); d >= 2
    auto origin = [&](int s, int& img, int& y0, int& x0) {
        const int k = per_img > 1 ? udiv((unsigned)s, (unsigned)per_img, p.m_img) : s;
        const int t = s - k * per_img, ty = p.tiles_w > 1 ? udiv((unsigned)t, (unsigned)p.tiles_w, p.m_tw) : t;
        img = by_image ? xcd + 8 * k : k;
        y0 = ty * DM_T; x0 = (t - ty * p.tiles_w) * DM_T;
    };

    // ---- zero what is read but never written: the plane columns 24 .. 47 ------------------------------------------------------------------
    for (int i = tid; i < (DM_LDS - DM_OFF_PLANAR) / 16; i += DM_THREADS) *reinterpret_cast<uint4*>(smem + DM_OFF_PLANAR + i * 16) = make_uint4(0, 0, 0, 0);

    // ---- Toeplitz fragments of this wave's 4 channels: element j of lane (n = li, k group lg) = T_ky[k = 8 lg + j][n] ------------------------
    // (the slab's 49 x 32 taps go through LDS once - coalesced loads, then branch-free selects; the region is the plane area zeroed above)
    float* s_taps = reinterpret_cast<float*>(smem + DM_OFF_PLANAR);         // [49][32] fp32, + [32] bias
    __syncthreads();
    for (int i = tid; i < 50 * DM_CB; i += DM_THREADS) {
        const int k = i / DM_CB, c = i - k * DM_CB;
        s_taps[i] = k < 49 ? p.w[(size_t)(FLIP ? 48 - k : k) * p.C + c0 + c] : (p.bias ? p.bias[c0 + c] : 0.f);
    }
    __syncthreads();
    bf16x8 T[CPW][7];
    float bia[CPW];
#pragma unroll
    for (int ch = 0; ch < CPW; ++ch) {
        const int c = CPW * wave + ch;
        bia[ch] = s_taps[49 * DM_CB + c];
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            unsigned pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = 8 * lg + 2 * e + h - li;                        // (flipped taps: tap index 48 - (ky * 7 + d) = (6 - ky) * 7 + (6 - d))
                    const int dc = d < 0 ? 0 : (d > 6 ? 6 : d);
                    const float t = s_taps[(ky * 7 + dc) * DM_CB + c];
                    v[h] = (d >= 0 && d <= 6) ? t : 0.f;
                }
                pk[e] = pack2bf(v[0], v[1]);
            }
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            T[ch][ky] = __builtin_bit_cast(bf16x8, (u32x4_t{pk[0], pk[1], pk[2], pk[3]}));
            __builtin_amdgcn_sched_barrier(0);          // (one fragment's eight LDS reads at a time: hoisted together they spill)
        }
    }
    __syncthreads();
    for (int i = tid; i < 50 * DM_CB / 4; i += DM_THREADS) *reinterpret_cast<uint4*>(smem + DM_OFF_PLANAR + i * 16) = make_uint4(0, 0, 0, 0);   // planes: zero again

    // ---- per-thread constants of the memory phases -------------------------------------------------------------------------------------------
    // Round 4, third version: the first two spent ~740 vector instructions per item and wave on index arithmetic (divisions by 22 and 6, 64-bit
    // address chains, bounds tests) around 28 MFMAs - tools/dwm_ablate.py: the "deposit + barriers" skeleton alone took a third of the kernel.
    // Everything that does not depend on the item is computed ONCE here; LDS addresses are one base register + immediates; a tile whose halo
    // lies inside the image (uniform test) takes global addresses "tile origin (scalar) + per-thread 32-bit offset" with no clamps / selects.
    // halo chunk it * 512 + tid -> pixel (hr, hc) of the 22 x 22 tile, channels 8 (tid & 3) ..
    const int part8 = (tid & 3) * 8;
    int h_off[NIN];                                               // element offset of the halo pixel from the tile's (y0 - 3, x0 - 3) corner
    // Thread id from the hardware, opaque to the optimiser (cnblock_bwdw.hip: bw_fresh_lane): what the EDGE tiles need per thread - halo row / column,
    // output pixel - is re-derived from it at its use.  Derived from `tid` it is loop-invariant, hipcc hoisted it, ran out of the 128 registers of a
    // 16-wave workgroup, and re-loaded it from scratch inside the item loop behind `s_waitcnt vmcnt(0)` - a drain of the halo prefetch per item.
    auto fresh_tid = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return wave * 64 + l;
    };
    auto halo_rc = [&](int it, int t, int& hr, int& hc) {
        const int pix = min((it * DM_THREADS + t) >> 2, DM_NPIX - 1);       // (chunks past the tile: a valid pixel again, written to the unused tail)
        hr = pix / DM_H; hc = pix - hr * DM_H;
    };
#pragma unroll
    for (int it = 0; it < NIN; ++it) {
        int hr, hc;
        halo_rc(it, tid, hr, hc);
        h_off[it] = (hr * p.W + hc) * p.C + part8;
    }
    int o_off[NOUT];                                               // output chunk it * 512 + tid -> pixel (row, xx) of the 16 x 16 tile
#pragma unroll
    for (int it = 0; it < NOUT; ++it) {
        const int pix = (it * DM_THREADS + tid) >> 2;
        o_off[it] = ((pix >> 4) * p.W + (pix & 15)) * p.C + part8;
    }
    // Every load of this kernel is UNCONDITIONAL and its result is first touched where it is consumed (round 4, second pass over this file): a
    // load under a per-lane condition, or a value zeroed right behind its load, is a use at the end of that branch - hipcc waited there with
    // vmcnt(0), i.e. for the whole two-items-ahead prefetch, in every edge tile (23 % of a 256 x 256 map) and, through the residual operand of
    // the data-gradient call, in every item.  Out-of-image halo pixels: clamped address now, zero at DEPOSIT time from a mask kept with the set.
    auto request = [&](uint4 (&set)[NIN], unsigned& inmask, int s) {
        int img, y0, x0;
        origin(s, img, y0, x0);
        const bf16_t* img_base = p.x + (size_t)img * p.H * p.W * p.C + c0;
        long el[NIN];
        unsigned m = 0xffffffffu;
        if (y0 >= 3 && x0 >= 3 && y0 + DM_T + 3 <= p.H && x0 + DM_T + 3 <= p.W) {          // (uniform) the halo lies inside the image
            const long corner = ((long)(y0 - 3) * p.W + (x0 - 3)) * p.C;
#pragma unroll
            for (int it = 0; it < NIN; ++it) el[it] = corner + h_off[it];
        } else {
            m = 0;
            const int ft = fresh_tid();
#pragma unroll
            for (int it = 0; it < NIN; ++it) {
                int hr, hc;
                halo_rc(it, ft, hr, hc);
                const int gy = y0 - 3 + hr, gx = x0 - 3 + hc;
                const bool in = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
                el[it] = ((long)cy * p.W + cx) * p.C + (ft & 3) * 8;
                m |= (in ? 1u : 0u) << it;
            }
        }
#pragma unroll
        for (int it = 0; it < NIN; ++it) set[it] = *reinterpret_cast<const uint4*>(img_base + el[it]);
        inmask = m;
    };
    char* dep_base = smem + DM_OFF_IN + tid * 16;
    auto deposit = [&](const uint4 (&set)[NIN], unsigned inmask) {              // registers -> NHWC LDS tile (pixels outside the image: zeros)
#pragma unroll
        for (int it = 0; it < NIN; ++it) {
            uint4 v = set[it];
            if (!((inmask >> it) & 1u)) v = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(dep_base + it * (DM_THREADS * 16)) = v;
        }
    };

    uint4 preA[NIN], preB[NIN];
#pragma unroll
    for (int it = 0; it < NIN; ++it) { preA[it] = make_uint4(0, 0, 0, 0); preB[it] = make_uint4(0, 0, 0, 0); }
    unsigned inA = 0xffffffffu, inB = 0xffffffffu;
    if (first < items) request(preA, inA, first);
    if (first + step < items) request(preB, inB, first + step);

    char* planar = smem + DM_OFF_PLANAR;
    // phase B: wave w moves halo rows w, w + NW, (w + 2 NW) (< 22): per row 12 units (6 groups of 4 columns x 2 channel halves) = 3 wave instructions
    constexpr int BR = (DM_H + NW - 1) / NW;                     // rows per wave (3 / 2)
    const int i16 = lane & 15, tq = i16 >> 2, tpc = i16 & 3;
    const char* b_src = smem + DM_OFF_IN + ((wave * DM_H + (lg >> 1) * 4 + tq) * 64 + (lg & 1) * 32 + tpc * 8);       // + r * NW * 22 * 64 + k * 512
    char* b_dst = planar + ((lg & 1) * 16 + i16) * DM_PLANE + wave * DM_ROWB + (lg >> 1) * 8;                          // + r * NW * 96 + k * 16
    // phase D
    char* d_plane = planar + (CPW * wave) * DM_PLANE;                                                                   // + ch * PLANE
    const int d_rd = li * DM_ROWB + lg * 16, d_wr = li * DM_OROWB + lg * 8;
    // phase E: wave w turns output rows ER w .. ER w + ER - 1; 16-lane group lg = 8-channel group
    constexpr int ER = DM_T / NW;
    const char* e_src = planar + (8 * lg + tq) * DM_PLANE + (ER * wave) * DM_OROWB + tpc * 8;                         // + it * 40 (+ 4 PLANE)
    char* e_dst = smem + DM_OFF_OUT + ((ER * wave) * DM_T + i16) * DM_OPIX + lg * 16;                                 // + it * 16 * 80
    // phase F
    const char* f_src = smem + DM_OFF_OUT + (tid >> 2) * DM_OPIX + (tid & 3) * 16;                                    // + it * (threads / 4) * 80

    auto one_item = [&](int item, uint4 (&pre)[NIN], unsigned& inm) {
        int img, y0, x0;
        origin(item, img, y0, x0);
        const size_t tile_el = ((size_t)img * p.H + y0) * p.W * p.C + (size_t)x0 * p.C + c0;      // (uniform) element offset of the tile's first pixel
        const bool full = y0 + DM_T <= p.H && x0 + DM_T <= p.W;                                     // (uniform) no output pixel outside the image
        // ================= A: this item's halo tile: registers -> LDS; the same registers then take item + 2 ============================
        deposit(pre, inm);
        // residual-gradient operand of the output pass (consumed in F, four barriers away) - requested BEFORE the halo tile of item + 2: the
        // vector-memory counter retires in order, so a wait for these registers in F also waits for every load issued before them.  With the
        // halo request first (round 4, first version) phase F drained the two-items-ahead prefetch every item: the data-gradient call ran 49 %
        // slower than the forward for 25 % more bytes.
        uint4 addv[NOUT];
        if constexpr (ADD) {
#pragma unroll
            for (int it = 0; it < NOUT; ++it) {              // (pixels past the image edge: a clamped address, never stored)
                size_t el = tile_el + o_off[it];
                if (!full) {                                   // (uniform)
                    const int ft = fresh_tid(), pix = (it * DM_THREADS + ft) >> 2;
                    const int py = min(y0 + (pix >> 4), p.H - 1), px = min(x0 + (pix & 15), p.W - 1);
                    el = ((size_t)img * p.H + py) * p.W * p.C + (size_t)px * p.C + c0 + (ft & 3) * 8;
                }
                addv[it] = *reinterpret_cast<const uint4*>(p.add + el);
            }
        }
        if (item + 2 * step < items && !(DWM_DBG & 16)) request(pre, inm, item + 2 * step);
        dm_barrier();
        // ================= B: NHWC halo tile -> channel planes (the LDS transposes: 4 pixels x 16 channels per 16 lanes) ================
        if constexpr (!(DWM_DBG & 1)) {
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                if (r + 1 < BR || wave < DM_H - (BR - 1) * NW) {  // (uniform: the last row group exists for the first waves only; EXEC stays full inside)
                    bf16x4 v[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) v[k] = dm_tr(b_src + r * (NW * DM_H * 64) + k * 512);
#pragma unroll
                    for (int k = 0; k < 3; ++k) *reinterpret_cast<bf16x4*>(b_dst + r * (NW * DM_ROWB) + k * 16) = v[k];
                }
            }
        }
        dm_barrier();
        // ================= D: 4 channels x 7 kernel rows; swapped operands: lane = row li, registers = x_out 4 lg .. 4 lg + 3 ============
        // Two channels at a time: their 14 fragments are requested together and the two accumulation chains alternate (a 16x16x32 MFMA
        // depends on its predecessor's accumulator).  A channel's outputs go back into the plane they were computed from: only this wave
        // touches its four planes in this phase and its LDS accesses stay in program order, so the writes follow the reads.
        if constexpr (!(DWM_DBG & 2)) {
#pragma unroll
            for (int cp = 0; cp < CPW / 2; ++cp) {
                char* pl0 = d_plane + (2 * cp) * DM_PLANE;
                // (fragments in two batches, ky 0-3 and 4-6: all 14 at once are 56 registers beside the 56 of T - with 16 waves (128 registers) hipcc
                //  spilled 26 and re-loaded per-item offsets from scratch INSIDE the item loop, each reload behind `s_waitcnt vmcnt(0)` = a drain of
                //  the two-items-ahead halo prefetch; round 4, ISA read)
                f32x4 acc0 = f32x4{bia[2 * cp], bia[2 * cp], bia[2 * cp], bia[2 * cp]};
                f32x4 acc1 = f32x4{bia[2 * cp + 1], bia[2 * cp + 1], bia[2 * cp + 1], bia[2 * cp + 1]};
#pragma unroll
                for (int kb = 0; kb < 7; kb += 4) {
                    bf16x8 a0[4], a1[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (kb + k < 7) {
                            a0[k] = *reinterpret_cast<const bf16x8*>(pl0 + d_rd + (kb + k) * DM_ROWB);
                            a1[k] = *reinterpret_cast<const bf16x8*>(pl0 + DM_PLANE + d_rd + (kb + k) * DM_ROWB);
                        }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (kb + k < 7) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T[2 * cp][kb + k], a0[k], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T[2 * cp + 1][kb + k], a1[k], acc1, 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                *reinterpret_cast<uint2*>(pl0 + d_wr) = make_uint2(pack2bf(acc0[0], acc0[1]), pack2bf(acc0[2], acc0[3]));
                *reinterpret_cast<uint2*>(pl0 + DM_PLANE + d_wr) = make_uint2(pack2bf(acc1[0], acc1[1]), pack2bf(acc1[2], acc1[3]));
                __builtin_amdgcn_sched_barrier(0);          // (two channels' fragments in flight at a time: 112 registers hold T)
            }
        }
        dm_barrier();
        // ================= E: output planes -> NHWC tile =================================================================================
        if constexpr (!(DWM_DBG & 4)) {
#pragma unroll
            for (int it = 0; it < ER; ++it) {
                const bf16x4 lo = dm_tr(e_src + it * DM_OROWB), hi = dm_tr(e_src + it * DM_OROWB + 4 * DM_PLANE);
                *reinterpret_cast<bf16x8*>(e_dst + it * (DM_T * DM_OPIX)) = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
        dm_barrier();
        // ================= F: 16-byte rows of the NHWC tile -> HBM (+ residual-gradient operand) =========================================
        // (no barrier after it: the next item's E, which re-writes the output tile, lies behind three more barriers)
        if constexpr (!(DWM_DBG & 8)) {
#pragma unroll
            for (int it = 0; it < NOUT; ++it) {
                const int pix = (it * DM_THREADS + tid) >> 2;
                if (full || (y0 + (pix >> 4) < p.H && x0 + (pix & 15) < p.W)) {
                    uint4 v = *reinterpret_cast<const uint4*>(f_src + it * ((DM_THREADS / 4) * DM_OPIX));
                    if constexpr (ADD) {
                        const uint4 a = addv[it];
                        v.x = pack2bf(bf2f_lo(v.x) + bf2f_lo(a.x), bf2f_hi(v.x) + bf2f_hi(a.x));
                        v.y = pack2bf(bf2f_lo(v.y) + bf2f_lo(a.y), bf2f_hi(v.y) + bf2f_hi(a.y));
                        v.z = pack2bf(bf2f_lo(v.z) + bf2f_lo(a.z), bf2f_hi(v.z) + bf2f_hi(a.z));
                        v.w = pack2bf(bf2f_lo(v.w) + bf2f_lo(a.w), bf2f_hi(v.w) + bf2f_hi(a.w));
                    }
                    store16_stream(p.y + tile_el + o_off[it], v, p.nt);
                }
            }
        }
    };
    for (int item = first; item < items; item += 2 * step) {
        one_item(item, preA, inA);
        if (item + step < items) one_item(item + step, preB, inB);
    }
}

MMG_API int mmg_dwconv7_nhwc_mfma(const void* x, const float* w, const float* bias, const void* add, void* y, int n, int H,
                                  int W, int C, int flip, hipStream_t stream) {
    MMG_CHECK_ARG(x && w && y, "mmg_dwconv7_nhwc_mfma: null pointer");
    MMG_CHECK_ARG(n > 0 && H > 0 && W > 0 && C > 0 && C % DM_CB == 0, "mmg_dwconv7_nhwc_mfma: n=%d H=%d W=%d C=%d (C must be a multiple of 32)", n, H, W, C);
    DwM p{(const bf16_t*)x, w, bias, (const bf16_t*)add, (bf16_t*)y, n, H, W, C, cdiv(W, DM_T), cdiv(H, DM_T), 0, 0, 0, 0};
    p.m_img = (unsigned)((1ULL << 32) / (unsigned long long)(p.tiles_w * p.tiles_h > 1 ? p.tiles_w * p.tiles_h : 2));
    p.m_tw = (unsigned)((1ULL << 32) / (unsigned long long)(p.tiles_w > 1 ? p.tiles_w : 2));
    p.nt = (size_t)n * H * W * C * 2 >= ((size_t)256 << 20);
    p.dbg = DWM_DBG;
    const long items = (long)n * p.tiles_w * p.tiles_h;
    const int slabs = C / DM_CB;
    MMG_CHECK_ARG(items < (1L << 30), "mmg_dwconv7_nhwc_mfma: too many tiles");
    // one workgroup per CU, a multiple of 8 (XCD groups); never fewer workgroups per XCD group than slabs
    int cus = mmg_cu_count_cached();
    int per_xcd = cus / 8;
    if (per_xcd < slabs) per_xcd = slabs;
    // small maps: no more streams than items
    const long want = (items + 7) / 8 * slabs;
    if (per_xcd > want) per_xcd = (int)(want < slabs ? slabs : want);
    const int grid = 8 * per_xcd;
    const int nw = getenv("MMG_DWM_WAVES") ? atoi(getenv("MMG_DWM_WAVES")) : 16;
    MMG_NOTE_KERNEL("dwconv7_mfma_kernel<%s, %d>", flip ? "true" : "false", nw == 8 ? 8 : 16);
#define DM_LAUNCH(FL, W_, AD)                                                                                 \
    do {                                                                                                      \
        mmg_allow_lds(dwconv7_mfma_kernel<FL, W_, AD>, DM_LDS);                                               \
        hipLaunchKernelGGL((dwconv7_mfma_kernel<FL, W_, AD>), dim3(grid), dim3(W_ * 64), DM_LDS, stream, p);  \
    } while (0)
#define DM_LAUNCH_W(FL, AD) do { if (nw == 8) DM_LAUNCH(FL, 8, AD); else DM_LAUNCH(FL, 16, AD); } while (0)
    if (flip) { if (add) DM_LAUNCH_W(true, true); else DM_LAUNCH_W(true, false); }
    else      { if (add) DM_LAUNCH_W(false, true); else DM_LAUNCH_W(false, false); }
#undef DM_LAUNCH_W
#undef DM_LAUNCH
    MMG_LAUNCH_CHECK("mmg_dwconv7_nhwc_mfma");
    return 0;
}
