// fhx_scan.hpp - ordered scan helpers shared by the sort-and-segment stages (fhx_kr.hip, fhx_cni.hip): run heads of a
// sorted u64 key array -> per-tile counts -> exclusive tile offsets.  Tiles of 1024 keys, 256 threads x 4 consecutive keys.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fhxscan {

constexpr int TILE = 1024;
constexpr int THREADS = 256;
constexpr int SCAN_ITEMS = 4;

// ordered block-wide exclusive scan of one small count per thread
static __device__ inline unsigned int block_exclusive_scan(unsigned int v, unsigned int* total) {
    __shared__ unsigned int wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned int inc = v;
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned int up = __shfl_up(inc, s, 64);
        if (lane >= s) inc += up;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned int base = 0, all = 0;
    for (int k = 0; k < 4; ++k) {
        if (k < w) base += wsum[k];
        all += wsum[k];
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

static __device__ inline bool is_head(const unsigned long long* keys, int64_t i) { return i == 0 || keys[i] != keys[i - 1]; }

static __global__ __launch_bounds__(THREADS) void count_heads(const unsigned long long* keys, int64_t N, unsigned int* tile_counts) {
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    unsigned int c = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < N && is_head(keys, base + k)) ++c;
    unsigned int total;
    block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

// exclusive scan of the tile counts (one block, any number of tiles); offsets are 64-bit
static __global__ __launch_bounds__(THREADS) void scan_tiles(const unsigned int* tile_counts, int64_t n_tiles, unsigned long long* tile_offsets,
                                                         unsigned long long* total_out) {
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_tiles; base += THREADS) {
        const int64_t i = base + threadIdx.x;
        const unsigned int v = i < n_tiles ? tile_counts[i] : 0;
        unsigned int total;
        const unsigned int ex = block_exclusive_scan(v, &total);
        if (i < n_tiles) tile_offsets[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

}  // namespace fhxscan
