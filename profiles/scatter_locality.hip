// scatter_locality.hip - what would a q scatter cost if the (row, q) pairs were grouped by row block first?
//   K3's bh_apply writes n_kept 8-byte values into a column of n_rows doubles at random rows: 0.50-0.55 ms for 1.5e7 of 1.2e8
//   (30 G stores/s, 4 x the algorithmic traffic: every store is a sector fill + a sector write-back at a random DRAM page).
//   Variants, same pairs:
//     random     : as bh_apply does today (nontemporal 8-byte stores in the order of the sorted p)
//     blocked    : pairs grouped into 256 row blocks (order inside a block random), plain stores, workgroups walk the array in order
//     blocked_xcd: the same, block b handled by the workgroups of XCD b % 8 only (workgroup id % 8 = XCD), one block at a time per XCD:
//                  a block's stretch of q (n_rows / 256 x 8 B) stays in that XCD's L2 while its stores arrive
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter_locality profiles/scatter_locality.hip && /tmp/scatter_locality [n_rows n_kept]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

__global__ __launch_bounds__(256) void k_random(const unsigned int* __restrict__ rows, const double* __restrict__ v, size_t n, double* __restrict__ q) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v[i], q + rows[i]);
}
__global__ __launch_bounds__(256) void k_plain(const unsigned int* __restrict__ rows, const double* __restrict__ v, size_t n, double* __restrict__ q) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;           // contiguous chunk per workgroup
    const size_t a = (size_t)blockIdx.x * per, b = a + per < n ? a + per : n;
    for (size_t i = a + threadIdx.x; i < b; i += 256) q[rows[i]] = v[i];
}
// block starts off[0..256]; XCD x = blockIdx.x % 8 takes blocks x, x + 8, ...; its gridDim.x / 8 workgroups share a block
__global__ __launch_bounds__(256) void k_xcd(const unsigned int* __restrict__ rows, const double* __restrict__ v, const size_t* __restrict__ off,
                                             double* __restrict__ q) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    for (int b = x; b < 256; b += 8) {
        const size_t a = off[b], e = off[b + 1];
        for (size_t i = a + (size_t)k * 256 + threadIdx.x; i < e; i += (size_t)per_xcd * 256) q[rows[i]] = v[i];
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t n_rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 120689521ull;
    const size_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 15086982ull;
    std::mt19937_64 rng(1);
    std::vector<unsigned int> rows(n);
    {   // n distinct-ish rows, random order (the order of the sorted p)
        for (size_t i = 0; i < n; ++i) rows[i] = (unsigned int)(rng() % n_rows);
    }
    std::vector<double> v(n, 0.25);
    // grouped by row block: block = row / ceil(n_rows / 256)
    const size_t bsz = (n_rows + 255) / 256;
    std::vector<size_t> off(257, 0);
    for (size_t i = 0; i < n; ++i) ++off[rows[i] / bsz + 1];
    for (int b = 0; b < 256; ++b) off[b + 1] += off[b];
    std::vector<unsigned int> grouped(n);
    {
        std::vector<size_t> at(off.begin(), off.end() - 1);
        for (size_t i = 0; i < n; ++i) grouped[at[rows[i] / bsz]++] = rows[i];
    }
    unsigned int *d_rows, *d_grouped;
    double *d_v, *d_q;
    size_t* d_off;
    hipMalloc(&d_rows, n * 4);
    hipMalloc(&d_grouped, n * 4);
    hipMalloc(&d_v, n * 8);
    hipMalloc(&d_q, n_rows * 8);
    hipMalloc(&d_off, 257 * sizeof(size_t));
    hipMemcpy(d_rows, rows.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_grouped, grouped.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_v, v.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_off, off.data(), 257 * sizeof(size_t), hipMemcpyHostToDevice);
    hipMemset(d_q, 0, n_rows * 8);
    std::printf("n_rows %zu (q column %.0f MB), %zu stores (%.1f %% of the rows), block = %zu rows = %.2f MB of q\n", n_rows, n_rows * 8 / 1e6, n,
                100.0 * n / n_rows, bsz, bsz * 8 / 1e6);
    const int grid = 256 * 8;
    const double t0 = time_ms([&] { hipLaunchKernelGGL(k_random, dim3(grid), dim3(256), 0, 0, d_rows, d_v, n, d_q); }, 10);
    const double t1 = time_ms([&] { hipLaunchKernelGGL(k_plain, dim3(grid), dim3(256), 0, 0, d_rows, d_v, n, d_q); }, 10);
    const double t2 = time_ms([&] { hipLaunchKernelGGL(k_plain, dim3(grid), dim3(256), 0, 0, d_grouped, d_v, n, d_q); }, 10);
    std::printf("random order, nontemporal 8-byte stores   %.3f ms\nrandom order, plain stores                %.3f ms\ngrouped by row block, plain, in order     %.3f ms\n", t0, t1, t2);
    for (int g : {256 * 2, 256 * 4, 256 * 8, 256 * 16}) {
        const double t3 = time_ms([&] { hipLaunchKernelGGL(k_xcd, dim3(g), dim3(256), 0, 0, d_grouped, d_v, d_off, d_q); }, 10);
        std::printf("grouped, one block per XCD at a time, %5d workgroups   %.3f ms\n", g, t3);
    }
    return 0;
}
