// Argument block shared by the two weight-gradient (TN) kernels: gemm_bf16.hip (128-wide tiles, any shape) and
// gemm_tn_wide.hip (192x384-class tiles, one 8-wave workgroup per CU, long reductions).
#pragma once
#include "common.h"

struct GemmTN {
    const bf16_t* A; const bf16_t* B;
    int M, N1, N2, lda, ldb;
    float* C; int ldc;
    float* colsum_a;          // [N1] += column sums of A (bias gradient of the same linear), nullable
    int tiles1, tiles2, rows_per_chunk;
    float alpha;
    int chunks, xcd_order;    // xcd_order: 0 = (tile, chunk) grid, 1 / 2 = XCD-grouped by the B / A block (see the kernels)
    // wide-tile kernel only: the caller's operands were exchanged (its A is this B) so that the wide side is N2; the tile is flushed
    // TRANSPOSED (C points at the caller's [N2, N1] matrix, ldc its leading dimension) and `colsum_a` holds sums of B's columns
    int swapped;
};

// Launches the wide-tile kernel when the shape suits it; returns false (nothing launched) otherwise.
bool mmg_tn_wide_launch(GemmTN& g, hipStream_t stream);
