// atomic_rate.hip - what an MI355X sustains in global atomic adds, by how many distinct addresses they spread over and whether the
// result is used: the number that decides whether K3's large sort can become one bucketing pass + an LDS sort per bucket
// (a counter per fine key bin: one atomic per key and pass) instead of eight radix passes.  1.6e7 operations per launch
// (the survivors of bench.py's k3_stress), addresses = a hash of the element index reduced to R words.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_rate profiles/atomic_rate.hip && /tmp/atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned int mix(unsigned int x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// keys of a radix / bucketing pass arrive in row order = random order of bins: address = hash(i) % r
template <bool RETURNING>
__global__ __launch_bounds__(256) void k_atomic(unsigned int* __restrict__ bins, unsigned int r, size_t n, unsigned int* __restrict__ sink) {
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned int b = mix((unsigned int)i) % r;
        if (RETURNING)
            acc += atomicAdd(&bins[b], 1u);
        else
            atomicAdd(&bins[b], 1u);               // result unused: the compiler emits the no-return form
    }
    if (RETURNING && acc == 0xdeadbeefu) sink[0] = acc;
}
// the same with the adds of a wave to one bin combined first (what helps when few bins are hot): leader lane per distinct bin
__global__ __launch_bounds__(256) void k_atomic_skewed(unsigned int* __restrict__ bins, unsigned int r, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // half of the keys in the top 1/16 of the bins (an exponent-like skew)
        const unsigned int h = mix((unsigned int)i);
        const unsigned int b = (h & 1u) ? (h >> 1) % (r / 16u + 1u) : (h >> 1) % r;
        atomicAdd(&bins[b], 1u);
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int k = 0; k < reps; ++k) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / reps;
}

int main() {
    const size_t n = 16000000;
    const unsigned int max_r = 1u << 25;
    unsigned int *bins = nullptr, *sink = nullptr;
    if (hipMalloc(&bins, (size_t)max_r * 4) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(bins, 0, (size_t)max_r * 4);
    const unsigned int rs[] = {1u, 64u, 4096u, 1u << 15, 1u << 18, 1u << 21, 1u << 23, 1u << 25};
    std::printf("%10s %14s %14s %14s   (ms per 1.6e7 atomic adds; G adds/s in brackets)\n", "addresses", "no return", "returning", "skewed, no ret");
    for (unsigned int r : rs) {
        const dim3 grid(256 * 8), block(256);
        const double a = time_ms([&] { hipLaunchKernelGGL(k_atomic<false>, grid, block, 0, 0, bins, r, n, sink); }, 3);
        const double b = time_ms([&] { hipLaunchKernelGGL(k_atomic<true>, grid, block, 0, 0, bins, r, n, sink); }, 3);
        const double c = time_ms([&] { hipLaunchKernelGGL(k_atomic_skewed, grid, block, 0, 0, bins, r, n); }, 3);
        std::printf("%10u %8.3f (%5.2f) %8.3f (%5.2f) %8.3f (%5.2f)\n", r, a, n / a * 1e-6, b, n / b * 1e-6, c, n / c * 1e-6);
    }
    return 0;
}
