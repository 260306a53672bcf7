#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void oob(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i <= n) p[i] = i; }   // writes p[n]: one past the end
int main() {
    int* d = nullptr;
    if (hipMalloc(&d, 256 * sizeof(int)) != hipSuccess) { std::printf("no device\n"); return 2; }
    hipLaunchKernelGGL(oob, dim3(2), dim3(256), 0, 0, d, 256);
    hipError_t e = hipDeviceSynchronize();
    std::printf("sync: %s\n", hipGetErrorString(e));
    return 0;
}
