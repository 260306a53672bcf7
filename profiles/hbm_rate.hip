// hbm_rate.hip - what HBM rates an MI355X sustains for the three access shapes of this path's streaming kernels, so that
// "per cent of the 8 TB/s spec" can be read next to "per cent of what a plain streaming kernel reaches":
//   read   : 16-byte loads, nothing stored (k1_classify_hist, the class kernels' queue reads)
//   write  : 16-byte stores, nothing loaded (a fill)
//   copy   : 16-byte load + 16-byte store per element pair (k3_compact: p in, q = 1 out - 1.18 GB each way on C3)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_rate profiles/hbm_rate.hip && /tmp/hbm_rate
// Sizes: 1.18 GB per array (C3's p column) and 4.7 GB; grids of 256 CUs x 8 workgroups like the engine's streaming kernels
// and "one workgroup per 4096 elements".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, size_t n2, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(double2* __restrict__ a, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x)
        a[i] = make_double2(1.0, 1.0);
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = a[i];
        b[i] = make_double2(v.x == v.x ? 1.0 : v.x, v.y == v.y ? 1.0 : v.y);
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / reps;
}

int main() {
    const size_t sizes[2] = {147964314ull * 8ull / 16ull * 16ull, 4ull * 147964314ull * 8ull / 16ull * 16ull};
    double* out = nullptr;
    hipMalloc(&out, 8);
    for (size_t bytes : sizes) {
        double2 *a = nullptr, *b = nullptr;
        if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) {
            std::printf("allocation of 2 x %.2f GB failed\n", bytes / 1e9);
            return 1;
        }
        hipMemset(a, 0, bytes);
        hipMemset(b, 0, bytes);
        const size_t n2 = bytes / 16;
        for (int grid : {2048, 4096, (int)((n2 + 2047) / 2048)}) {
            const double r = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n2, out); }, 10);
            const double w = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n2); }, 10);
            const double c = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n2); }, 10);
            std::printf("%.2f GB per array, grid %7d x 256: read %.3f ms = %.2f TB/s   write %.3f ms = %.2f TB/s   copy %.3f ms = %.2f TB/s "
                        "(both directions counted)\n",
                        bytes / 1e9, grid, r, bytes / r / 1e9, w, bytes / w / 1e9, c, 2.0 * bytes / c / 1e9);
        }
        const double m = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 10);
        std::printf("%.2f GB per array, hipMemcpyAsync device to device: %.3f ms = %.2f TB/s (both directions counted)\n", bytes / 1e9, m,
                    2.0 * bytes / m / 1e9);
        hipFree(a);
        hipFree(b);
    }
    return 0;
}
