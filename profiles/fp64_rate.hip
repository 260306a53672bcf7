// fp64_rate.hip - what rate of fp64 vector instructions an MI355X sustains (the ceiling k2h_heavy is priced against).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_rate profiles/fp64_rate.hip && /tmp/fp64_rate
// Nominal: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3e12 fp64 instruction-lanes/s (78.6 TFLOP/s counting an FMA as 2).
// Variants: CHAINS independent dependency chains per lane (1 = pure latency chain, 8 = issue-bound), fma or mul+add pairs,
// WAVES waves per SIMD; each launch runs long enough (tens of ms) for the clocks to settle.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int CHAINS, bool FMA>
__global__ __launch_bounds__(256) void spin(double* out, double b, double c, int iters) {
    double a[CHAINS];
    for (int k = 0; k < CHAINS; ++k) a[k] = 1.0 + 1e-9 * (threadIdx.x + k);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) {
            if (FMA) {
                a[k] = __builtin_fma(a[k], b, c);
            } else {
                a[k] = a[k] * b;
                a[k] = a[k] + c;
            }
        }
    }
    double s = 0;
    for (int k = 0; k < CHAINS; ++k) s += a[k];
    if (s == 12345.678) out[0] = s;                      // never true: keeps the loop
}

template <int CHAINS, bool FMA>
void run(const char* label, int waves_per_simd, int iters, double* d_out) {
    const int blocks = 256 * waves_per_simd;              // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((spin<CHAINS, FMA>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001, 1e-12, iters / 10);   // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((spin<CHAINS, FMA>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001, 1e-12, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 256 * (double)iters * CHAINS * (FMA ? 1 : 2);     // instruction-lanes
    std::printf("%-44s %2d waves/SIMD: %8.2f ms, %6.2f e12 instr-lanes/s = %5.1f %% of nominal\n", label, waves_per_simd, ms, instr / ms / 1e9,
                100.0 * instr / (ms * 1e-3) / 39.3216e12);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    double* d_out = nullptr;
    hipMalloc(&d_out, 8);
    const int it = 400000;
    for (int w : {1, 2, 4, 8}) run<8, true>("v_fma_f64, 8 independent chains", w, it, d_out);
    for (int w : {1, 2, 4, 8}) run<1, true>("v_fma_f64, one dependent chain", w, it * 4, d_out);
    for (int w : {1, 2, 4, 8}) run<1, false>("v_mul_f64 + v_add_f64, one dependent chain", w, it * 2, d_out);
    for (int w : {4, 8}) run<8, false>("v_mul_f64 + v_add_f64, 8 chains", w, it / 2, d_out);
    hipFree(d_out);
    return 0;
}
