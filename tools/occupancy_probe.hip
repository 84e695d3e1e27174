// How many workgroups of 320 threads fit a gfx950 compute unit for a given static LDS size?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB>
__global__ void __launch_bounds__(320) k_lds(float *out) {
    __shared__ float buf[KB * 256];
    buf[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (out) out[threadIdx.x] = buf[(threadIdx.x * 7) % (KB * 256)];
}
template <int KB> void probe() {
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_lds<KB>, 320, 0);
    printf("static LDS %3d KB, 320 threads: %d workgroups per CU (%s)\n", KB, n, hipGetErrorString(e));
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  regsPerBlock %d\n", p.sharedMemPerBlock,
           p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
    probe<32>(); probe<48>(); probe<52>(); probe<56>(); probe<60>(); probe<64>(); probe<80>(); probe<128>(); probe<160>();
    return 0;
}
