// What does ds_read_b64_tr_b8 deliver on gfx950?  Every lane supplies the LDS byte address lane * 8 (+ base); the LDS holds byte id = address.
// Output: for every lane its 8 result bytes as (source lane, byte within that lane's 8) pairs - the lane / byte permutation of the instruction.
// Build: hipcc -O3 --offload-arch=gfx950 -o tr_b8_probe tr_b8_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
__global__ void probe(unsigned* out_lo, unsigned* out_hi) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[1024];
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = pass == 0 ? (unsigned char)(i & 255) : (unsigned char)(i >> 8);
        __syncthreads();
        i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(buf + threadIdx.x * 8));
        unsigned* o = pass == 0 ? out_lo : out_hi;
        o[threadIdx.x * 2] = (unsigned)v[0]; o[threadIdx.x * 2 + 1] = (unsigned)v[1];
        __syncthreads();
    }
}
int main() {
    unsigned *lo, *hi, hlo[128], hhi[128];
    hipMalloc(&lo, 512); hipMalloc(&hi, 512);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, lo, hi);
    hipMemcpy(hlo, lo, 512, hipMemcpyDeviceToHost); hipMemcpy(hhi, hi, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 8; ++j) {
            const unsigned a = ((hlo[l * 2 + j / 4] >> (8 * (j % 4))) & 255) | (((hhi[l * 2 + j / 4] >> (8 * (j % 4))) & 255) << 8);
            printf("  (L%2u,b%u)", a / 8, a % 8);
        }
        printf("\n");
    }
    return 0;
}
