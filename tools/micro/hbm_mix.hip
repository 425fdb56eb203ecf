// What HBM rate do loads, stores and mixes of the two reach on gfx950, by access shape?  Round 3 read the NT GEMM's dgelu epilogue and the fused
// CNBlock backward as bound by "the write path" (~3.4 TB/s) from kernels that also do other work; this probe measures the memory system alone.
//   mode R  : every lane loads 16 B (sum kept alive), nothing stored            mode W  : every lane stores 16 B, nothing loaded
//   mode C  : copy (1 load : 1 store)                                            mode RW2: 1 load : 2 stores (the dgelu epilogue: hpre in, dh + g out)
//   store kinds: plain global_store_dwordx4, the same with nt, buffer_store
//   shape: a wave instruction covers `seg` contiguous bytes of one row and 1024/seg rows `stride` bytes apart (seg = 1024: fully linear;
//          seg = 512, stride = 3072: the 256-column tile of a [M,1536] bf16 matrix, i.e. the epilogue's stores)
// Build: hipcc -O3 --offload-arch=gfx950 -o hbm_mix hbm_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct Args {
    const u32x4* in; u32x4* out0; u32x4* out1;
    long long rows;        // rows of `stride` bytes
    int stride;            // bytes per row
    int seg;               // contiguous bytes a workgroup column-tile covers per row (multiple of 16, divides 1024 or is a multiple of it)
};

// A workgroup of 256 threads walks a tile of TR rows x seg bytes (TR = 16 KB / seg rows: up to four 16-B granules per lane) and then the next tile
// (grid-stride over tiles; tiles are numbered column-tile fastest, as a GEMM's N-tiles would be).
template <int MODE, int ST>
__global__ __launch_bounds__(256) void mix_kernel(const Args a) {
    const int segs = a.stride / a.seg;                    // column tiles per row
    const int gpr = a.seg / 16;                           // 16-B granules per row segment
    const int TR = 16384 / a.seg;                         // rows per tile
    const long long tiles = (a.rows / TR) * segs;
    u32x4 acc = {0, 0, 0, 0};
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long long r0 = (t / segs) * TR; const int c0 = (int)(t % segs) * a.seg;
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = threadIdx.x + i * 256; const int r = g / gpr, c = (g % gpr) * 16;
            if (g >= TR * gpr) continue;
            const long long off = ((r0 + r) * (long long)a.stride + c0 + c) / 16;
            if (MODE != 1) v[i] = __builtin_nontemporal_load(a.in + off); else v[i] = u32x4{(unsigned)g, (unsigned)t, 1u, 2u};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = threadIdx.x + i * 256; const int r = g / gpr, c = (g % gpr) * 16;
            if (g >= TR * gpr) continue;
            const long long off = ((r0 + r) * (long long)a.stride + c0 + c) / 16;
            if (MODE == 0) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
            else {
                if (ST == 0) a.out0[off] = v[i]; else __builtin_nontemporal_store(v[i], a.out0 + off);
                if (MODE == 3) { u32x4 w = v[i]; w.x ^= 0x5555u; if (ST == 0) a.out1[off] = w; else __builtin_nontemporal_store(w, a.out1 + off); }
            }
        }
    }
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) a.out0[0] = acc;
}

template <int MODE, int ST>
static void run(const char* name, Args a, int grid, double bytes_per_launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((mix_kernel<MODE, ST>), dim3(grid), dim3(256), 0, 0, a);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mix_kernel<MODE, ST>), dim3(grid), dim3(256), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("  %-28s %8.3f ms   %6.2f TB/s   %6.1f B/cycle/workgroup at 1.85 GHz\n", name, ms, bytes_per_launch / ms * 1e-9, bytes_per_launch / (ms * 1e-3) / grid / 1.85e9);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    const long long bytes = 3221225472LL;                                  // 3 GiB per stream: well past the 256 MB infinity cache
    u32x4 *in, *o0, *o1;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&o0, bytes) != hipSuccess || hipMalloc(&o1, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 1, bytes); hipMemset(o0, 0, bytes); hipMemset(o1, 0, bytes);
    struct Shape { int stride, seg; const char* what; } shapes[] = {
        {1024, 1024, "linear (seg = stride = 1 KB)"},
        {3072, 3072, "rows of 1536 bf16, whole rows"},
        {3072, 512, "rows of 1536 bf16, 256-column tiles (NT epilogue)"},
        {3072, 256, "rows of 1536 bf16, 128-column tiles"},
        {768, 768, "rows of 384 bf16, whole rows"},
        {192, 192, "rows of 96 bf16, whole rows"},
        {6144, 512, "rows of 3072 bf16, 256-column tiles"},
    };
    // argv: grids to run instead of the two defaults (e.g. "32 64 128 256": what ONE CU can push when HBM is not the limit)
    int grids[8] = {256 * 2, 256 * 8, 0, 0, 0, 0, 0, 0};
    int ngrids = 2;
    if (argc > 1) { ngrids = 0; for (int i = 1; i < argc && ngrids < 8; ++i) grids[ngrids++] = atoi(argv[i]); }
    for (const Shape& s : shapes) {
        Args a{in, o0, o1, bytes / s.stride, s.stride, s.seg};
        a.rows -= a.rows % (16384 / s.seg > 0 ? 16384 / s.seg : 1);
        const double b = (double)a.rows * s.stride;
        for (int gi = 0; gi < ngrids; ++gi) {
            const int grid = grids[gi];
            printf("%s, %d workgroups of 256\n", s.what, grid);
            run<0, 0>("read only", a, grid, b);
            run<1, 0>("write only", a, grid, b);
            run<1, 1>("write only, nt", a, grid, b);
            run<2, 0>("copy 1R:1W", a, grid, 2 * b);
            run<2, 1>("copy 1R:1W, nt stores", a, grid, 2 * b);
            run<3, 0>("1R:2W", a, grid, 3 * b);
            run<3, 1>("1R:2W, nt stores", a, grid, 3 * b);
        }
    }
    hipFree(in); hipFree(o0); hipFree(o1);
    return 0;
}
