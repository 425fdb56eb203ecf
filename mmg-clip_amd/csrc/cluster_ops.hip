// AveragedMedicalCLIPLoss on the device (reference mmgclip/loss/losses.py:98-216): greedy threshold clustering of the text
// similarity matrix (`_assign_labels`, :148-162: a Python double loop with one device read per element in the reference) and
// the per-cluster column mean of the logits (`_average_logits`, :164-186) with its backward.
#include "common.h"

// One workgroup.  Leaders are found in index order exactly as the reference does: text i still unlabelled opens cluster
// `cur`, every still unlabelled j > i with sim[i][j] >= threshold joins it.  The scan over j is parallel, the walk over i is
// sequential with one barrier per LEADER (not per row): labels live in LDS.
__global__ __launch_bounds__(1024) void greedy_threshold_labels_kernel(const float* __restrict__ sim, int ld, int n, float thr,
                                                                       long long* __restrict__ labels, int* __restrict__ counts,
                                                                       int* __restrict__ k_out) {
    extern __shared__ int lab[];
    const int tid = threadIdx.x;
    for (int j = tid; j < n; j += 1024) { lab[j] = -1; counts[j] = 0; }
    __syncthreads();
    int cur = 0;
    for (int i = 0; i < n; ++i) {
        if (lab[i] != -1) continue;            // the same LDS word for every thread: uniform branch
        __syncthreads();                       // everybody has read lab[i] before thread 0 overwrites it
        const float* row = sim + (size_t)i * ld;
        for (int j = i + 1 + tid; j < n; j += 1024)
            if (lab[j] == -1 && row[j] >= thr) lab[j] = cur;
        if (tid == 0) lab[i] = cur;
        ++cur;
        __syncthreads();
    }
    for (int j = tid; j < n; j += 1024) {
        labels[j] = lab[j];
        atomicAdd(&counts[lab[j]], 1);
    }
    if (tid == 0) *k_out = cur;
}

// out[i, c] = mean over {j : labels[j] == c} of logits[i, j]; one workgroup per row
__global__ __launch_bounds__(256) void cluster_mean_cols_fwd_kernel(const float* __restrict__ logits, int ld, int N,
                                                                    const long long* __restrict__ labels,
                                                                    const int* __restrict__ counts, int k,
                                                                    float* __restrict__ out, int ldo) {
    extern __shared__ float acc[];
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < k; c += 256) acc[c] = 0.f;
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += 256) atomicAdd(&acc[(int)labels[j]], logits[(size_t)i * ld + j]);
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += 256) out[(size_t)i * ldo + c] = acc[c] / (float)counts[c];
}

// dlogits[i, j] = dout[i, labels[j]] / counts[labels[j]]
__global__ __launch_bounds__(256) void cluster_mean_cols_bwd_kernel(const float* __restrict__ dout, int ldo, int n, int N,
                                                                    const long long* __restrict__ labels,
                                                                    const int* __restrict__ counts,
                                                                    float* __restrict__ dlogits, int ld) {
    const size_t total = (size_t)n * N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int i = (int)(idx / N), j = (int)(idx % N);
        const int c = (int)labels[j];
        dlogits[(size_t)i * ld + j] = dout[(size_t)i * ldo + c] / (float)counts[c];
    }
}

MMG_API int mmg_greedy_threshold_labels(const float* sim, int ld, int n, float threshold, long long* labels, int* counts,
                                        int* k_out, hipStream_t stream) {
    MMG_CHECK_ARG(sim && labels && counts && k_out && n > 0 && n <= 16384 && ld >= n,
                  "mmg_greedy_threshold_labels: n=%d (1..16384) ld=%d", n, ld);
    mmg_allow_lds(greedy_threshold_labels_kernel, (size_t)n * sizeof(int));
    hipLaunchKernelGGL(greedy_threshold_labels_kernel, dim3(1), dim3(1024), (size_t)n * sizeof(int), stream, sim, ld, n, threshold,
                       labels, counts, k_out);
    MMG_LAUNCH_CHECK("mmg_greedy_threshold_labels");
    return 0;
}

MMG_API int mmg_cluster_mean_cols_fwd(const float* logits, int ld, int n, int N, const long long* labels, const int* counts,
                                      int k, float* out, int ldo, hipStream_t stream) {
    MMG_CHECK_ARG(logits && labels && counts && out && n > 0 && N > 0 && k > 0 && k <= N && k <= 16384 && ld >= N && ldo >= k,
                  "mmg_cluster_mean_cols_fwd: n=%d N=%d k=%d ld=%d ldo=%d", n, N, k, ld, ldo);
    mmg_allow_lds(cluster_mean_cols_fwd_kernel, (size_t)k * sizeof(float));
    hipLaunchKernelGGL(cluster_mean_cols_fwd_kernel, dim3(n), dim3(256), (size_t)k * sizeof(float), stream, logits, ld, N, labels,
                       counts, k, out, ldo);
    MMG_LAUNCH_CHECK("mmg_cluster_mean_cols_fwd");
    return 0;
}

MMG_API int mmg_cluster_mean_cols_bwd(const float* dout, int ldo, int n, int N, const long long* labels, const int* counts,
                                      float* dlogits, int ld, hipStream_t stream) {
    MMG_CHECK_ARG(dout && labels && counts && dlogits && n > 0 && N > 0 && ld >= N, "mmg_cluster_mean_cols_bwd: bad argument");
    const size_t total = (size_t)n * N;
    int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(cluster_mean_cols_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dout, ldo, n, N, labels, counts, dlogits, ld);
    MMG_LAUNCH_CHECK("mmg_cluster_mean_cols_bwd");
    return 0;
}
