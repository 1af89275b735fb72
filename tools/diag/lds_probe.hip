// Diagnostic: LDS limits on the box and whether large dynamic LDS launches work.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int n) {
    extern __shared__ int s[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = i;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[n - 1];
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("name %s arch %s CUs %d sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu optin %zu regsPerBlock %d clock %d\n", p.name, p.gcnArchName,
           p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin, p.regsPerBlock, p.clockRate);
    int* d; hipMalloc(&d, 1024);
    for (int kb : {32, 64, 65, 96, 128, 159, 160}) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        printf("setattr %d KB -> %s\n", kb, hipGetErrorString(e));
        hipLaunchKernelGGL(k, dim3(4), dim3(256), kb * 1024, 0, d, kb * 256);
        hipError_t l = hipGetLastError(); hipError_t s = hipDeviceSynchronize();
        int h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("  launch %d KB -> %s / %s  out=%d (expect %d)\n", kb, hipGetErrorString(l), hipGetErrorString(s), h, kb * 256 - 1);
    }
    return 0;
}
