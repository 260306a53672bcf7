// fhx_kr.hip - Knight-Ruiz matrix balancing on MI355X (gfx950): the bias-vector step that precedes Fit-Hi-C
// (reference: fithic/utils/HiCKRy.py).
//
//   assemble   HiCKRy.py:18-54    (chr,mid) -> locus index by binary search, keys row*n+col of the rows and of their
//                                 transposes, stable radix sort, duplicates added one by one in file order -> CSR
//   sparse     HiCKRy.py:74-101   row sums (SpMV with ones), threshold on the host (n values), row/column compaction
//   balance    HiCKRy.py:139-243  Newton / conjugate-gradient iteration; one SpMV per inner step, everything else is
//                                 n-sized vector work; the control flow (thresholds, breaks) runs on the host on scalars
//   bias       HiCKRy.py:103-115  (1/x) / mean(1/x), -1 at the removed rows
//
// HBM layout: CSR with int64 indptr, int32 column, double value (12 B per stored cell); all vectors double.
// The dominant kernel is kr_spmv: one wave64 per row, four consecutive cells per lane (16-byte column and value loads, a wave
// reads contiguous KBs), the input vector is gathered (it is n*8 bytes, L2/MALL resident), partial sums meet in a shuffle tree.
// It is HBM bound: 12 B per stored cell + 8 B per row gathered/written.  blockIdx -> row mapping gives every XCD one
// contiguous band of rows so that the gathers of neighbouring rows hit the same L2.
//
// Summation orders are fixed (and restated in oracle/kr_oracle.c, which the tests compare bit for bit):
//   row sum      chunks of 256 cells from the row start; lane l adds cells 4l..4l+3 of every chunk in cell order, chunks in
//                order; then v[l] += v[l+s] for s = 32..1
//   dot / sum    tiles of 1024 elements: thread t adds elements t, t+256, t+512, t+768; wave tree; the four waves and
//                then the tiles are added left to right
// The reference's own last bits depend on its BLAS (ddot) build, so this is the parity that can be stated: equal to
// the oracle bit for bit, and to the reference's vectors within 1e-12 relative on the committed goldens.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fithic_mi355x.h"
#include "fhx_scan.hpp"

namespace krd {

constexpr int TILE = 1024;          // elements per reduction tile
constexpr int THREADS = 256;
constexpr int SCAN_ITEMS = 4;       // consecutive elements per thread in the ordered scans

__device__ inline double wave_tree_sum(double v) {
    for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_down(v, s, 64);
    return v;                       // lane 0: v[l] += v[l+s] tree
}

__device__ inline double nan_min(double a, double b) { return (a < b || a != a) ? a : b; }   // NaN wins, like np.amin
__device__ inline double nan_max(double a, double b) { return (a > b || a != a) ? a : b; }

__device__ inline double wave_min(double v) {
    for (int s = 32; s >= 1; s >>= 1) v = nan_min(v, __shfl_down(v, s, 64));
    return v;
}
__device__ inline double wave_max(double v) {
    for (int s = 32; s >= 1; s >>= 1) v = nan_max(v, __shfl_down(v, s, 64));
    return v;
}

// one partial per tile; op 0 = ordered sum, 1 = min, 2 = max
template <int OP>
__device__ inline void tile_reduce_store(double acc, double* partials) {
    __shared__ double wpart[3][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double r = OP == 0 ? wave_tree_sum(acc) : (OP == 1 ? wave_min(acc) : wave_max(acc));
    if (lane == 0) wpart[OP][w] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = wpart[OP][0];
        for (int k = 1; k < 4; ++k) t = OP == 0 ? t + wpart[OP][k] : (OP == 1 ? nan_min(t, wpart[OP][k]) : nan_max(t, wpart[OP][k]));
        partials[blockIdx.x] = t;
    }
}

// final pass over the tile partials: strictly left to right for sums (one thread; loads staged through LDS in blocks
// of 8 so that only the adds are serial)
template <int OP>
__global__ __launch_bounds__(THREADS) void kr_finish(const double* partials, int64_t n_tiles, double* out) {
    __shared__ double buf[TILE];
    double total = 0.0;
    for (int64_t base = 0; base < n_tiles; base += TILE) {
        const int len = (int)min((int64_t)TILE, n_tiles - base);
        for (int i = threadIdx.x; i < len; i += THREADS) buf[i] = partials[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            int i = 0;
            if (base == 0) {
                total = buf[0];
                i = 1;
            }
            auto fold = [](double t, double v) { return OP == 0 ? t + v : (OP == 1 ? nan_min(t, v) : nan_max(t, v)); };
            for (; i + 8 <= len; i += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = buf[i + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) total = fold(total, v[u]);
            }
            for (; i < len; ++i) total = fold(total, buf[i]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = total;
}

// ---------------------------------------------------------------------------------------------------------------
// SpMV: one wave per row.  MODE 0: out0 = A in;  1: out0 = a0*(A in), out1 = 1 - out0  (v, rk; HiCKRy.py:153-154,222-223)
//                          2: out0 = a0*(A in) + a1*a2                                   (w; HiCKRy.py:192)
// A row is cut into chunks of 256 cells; lane l owns cells 4l..4l+3 of every chunk and adds them in cell order, chunks in
// order: one 16-byte column load and one 16-byte (binary32) / two 16-byte (double) value loads per lane and chunk instead
// of four 4-byte ones (a row of ~500 cells = two chunks: four loads in flight, then eight gathers).  The loads are only dword aligned (rows start anywhere); a wave's 64 loads still cover one contiguous KB.
// The last chunk reads past the row (the head of the next row - the next wave's data, so not wasted; the arrays carry 256
// cells of padding behind the last row) and predicates the adds.
// ---------------------------------------------------------------------------------------------------------------
typedef int kr_i4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float kr_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef double kr_d2 __attribute__((ext_vector_type(2), aligned(8)));
constexpr int KR_CHUNK = 256;       // cells per wave step
constexpr int KR_PAD = 256;         // cells allocated behind the last row of col / val

template <typename VT>
struct KrCells;
template <>
struct KrCells<float> {             // 8 VGPRs per chunk: the values stay binary32 until they are used
    kr_i4 c;
    kr_f4 f;
    __device__ __forceinline__ double v(int i) const { return (double)(i == 0 ? f.x : (i == 1 ? f.y : (i == 2 ? f.z : f.w))); }
};
template <>
struct KrCells<double> {
    kr_i4 c;
    kr_d2 a, b;
    __device__ __forceinline__ double v(int i) const { return i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? b.x : b.y)); }
};
__device__ __forceinline__ KrCells<float> kr_load_cells(const int32_t* __restrict__ col, const float* __restrict__ val, int64_t j) {
    KrCells<float> k;
    k.c = *reinterpret_cast<const kr_i4*>(col + j);
    k.f = *reinterpret_cast<const kr_f4*>(val + j);
    return k;
}
__device__ __forceinline__ KrCells<double> kr_load_cells(const int32_t* __restrict__ col, const double* __restrict__ val, int64_t j) {
    KrCells<double> k;
    k.c = *reinterpret_cast<const kr_i4*>(col + j);
    k.a = *reinterpret_cast<const kr_d2*>(val + j);
    k.b = *reinterpret_cast<const kr_d2*>(val + j + 2);
    return k;
}

// one chunk into the row's partial: cells past the row end gather x[0] (the four gathers go out together, no branch) and
// skip the add
template <typename VT>
__device__ __forceinline__ void kr_gather(const KrCells<VT>& k, int64_t j, int64_t e, const double* __restrict__ in, double (&x)[4]) {
    x[0] = in[j < e ? k.c.x : 0];
    x[1] = in[j + 1 < e ? k.c.y : 0];
    x[2] = in[j + 2 < e ? k.c.z : 0];
    x[3] = in[j + 3 < e ? k.c.w : 0];
}
template <typename VT>
__device__ __forceinline__ double kr_fold(double acc, const KrCells<VT>& k, const double (&x)[4], int64_t j, int64_t e) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double s = acc + k.v(i) * x[i];
        acc = j + i < e ? s : acc;
    }
    return acc;
}

// Measured on the C3 matrix (2.96e8 cells, 547 332 rows; profiles/history/r02_n_kr_variants.txt, r02_o_kr_variants.txt): this form
// 0.44-0.47 ms = 4.8-5.1 TB/s (the streaming read ceiling seen on this part; K1 reaches 5.2); 4-byte loads with lane-strided
// cells 0.50; a wave walking 4 / 8 / 16 rows as a software pipeline (next row's cells and the bounds of the row after it
// requested before the current row is reduced) 0.56-0.58 - the kernel is not latency bound; cells transposed through LDS so that
// each gather instruction covers 64 consecutive cells 0.47 - nor bound by the number of lines a gather touches.
template <int MODE, typename VT>
__global__ __launch_bounds__(THREADS) void kr_spmv(int64_t n, const int64_t* __restrict__ indptr, const int32_t* __restrict__ col,
                                                   const VT* __restrict__ val, const double* __restrict__ in,
                                                   double* __restrict__ out0, double* __restrict__ out1,
                                                   const double* __restrict__ a0, const double* __restrict__ a1,
                                                   const double* __restrict__ a2, int64_t n_blocks) {
    // XCD-aware: workgroup b runs on XCD b % 8; give each XCD a contiguous band of row blocks
    const int64_t per = (n_blocks + 7) / 8;
    const int64_t blk = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (blk >= n_blocks) return;
    const int lane = threadIdx.x & 63;
    const int64_t row = blk * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t b = indptr[row], e = indptr[row + 1];
    double acc = 0.0;
    for (int64_t base = b; base < e; base += 2 * KR_CHUNK) {           // two chunks per step: all four loads go out first
        const int64_t j0 = base + 4 * lane, j1 = j0 + KR_CHUNK;
        const bool second = base + KR_CHUNK < e;                        // wave-uniform
        const KrCells<VT> k0 = kr_load_cells(col, val, j0);
        KrCells<VT> k1 = k0;
        if (second) k1 = kr_load_cells(col, val, j1);
        double x0[4], x1[4];
        kr_gather(k0, j0, e, in, x0);
        if (second) kr_gather(k1, j1, e, in, x1);
        acc = kr_fold(acc, k0, x0, j0, e);
        if (second) acc = kr_fold(acc, k1, x1, j1, e);
    }
    const double t = wave_tree_sum(acc);
    if (lane == 0) {
        if (MODE == 0) {
            out0[row] = t;
        } else if (MODE == 1) {
            const double v = a0[row] * t;
            out0[row] = v;
            out1[row] = 1.0 - v;
        } else {
            out0[row] = a0[row] * t + a1[row] * a2[row];
        }
    }
}

// Contact counts are integers: when every stored value is exactly representable in binary32 the matrix is streamed as
// float (8 B per cell instead of 12) and widened in the kernel - the same doubles enter the same products.
__global__ __launch_bounds__(THREADS) void kr_to_f32(const double* __restrict__ val, int64_t nnz, float* __restrict__ out,
                                                     unsigned int* __restrict__ inexact) {
    unsigned int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = val[i];
        const float f = (float)v;
        out[i] = f;
        if (!((double)f == v)) bad = 1;                    // also catches NaN
    }
    if (__ballot(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(inexact, 1u);
}

// The 4-byte cell.  A Hi-C map at one resolution is a band: the columns of a row lie within a few hundred of each other, and
// the counts are small integers.  When every row spans fewer than 65 536 columns and every value is an integer below 65 536
// the matrix is streamed as one dword per cell - (column - first column of the row) | value << 16 - plus the row's first
// column: 4 B per cell instead of 8 (binary32 values) or 12.  The same doubles enter the same products in the same order, so
// the results are the same bits; any row or value that does not fit leaves the matrix in the wider format.
typedef unsigned int kr_u4 __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(THREADS) void kr_pack16(int64_t n, const int64_t* __restrict__ indptr, const int32_t* __restrict__ col,
                                                     const double* __restrict__ val, unsigned int* __restrict__ pack,
                                                     int32_t* __restrict__ rowbase, unsigned int* __restrict__ misfit) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t b = indptr[row], e = indptr[row + 1];
    int lo = 0x7fffffff;
    for (int64_t j = b + lane; j < e; j += 64) lo = min(lo, col[j]);
    for (int s = 32; s >= 1; s >>= 1) lo = min(lo, __shfl_xor(lo, s, 64));
    if (e == b) lo = 0;
    if (lane == 0) rowbase[row] = lo;
    bool bad = false;
    for (int64_t j = b + lane; j < e; j += 64) {
        const long long off = (long long)col[j] - lo;
        const double v = val[j];
        const unsigned int q = v >= 0.0 && v < 65536.0 ? (unsigned int)v : 0u;
        bad |= off > 65535 || !((double)q == v);
        pack[j] = (unsigned int)(off & 0xFFFF) | (q << 16);
    }
    if (__ballot(bad) && lane == 0) atomicOr(misfit, 1u);
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void kr_spmv_packed(int64_t n, const int64_t* __restrict__ indptr, const unsigned int* __restrict__ pack,
                                                          const int32_t* __restrict__ rowbase, const double* __restrict__ in,
                                                          double* __restrict__ out0, double* __restrict__ out1, const double* __restrict__ a0,
                                                          const double* __restrict__ a1, const double* __restrict__ a2, int64_t n_blocks) {
    const int64_t per = (n_blocks + 7) / 8;                             // XCD-aware, as kr_spmv
    const int64_t blk = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (blk >= n_blocks) return;
    const int lane = threadIdx.x & 63;
    const int64_t row = blk * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int64_t b = indptr[row], e = indptr[row + 1];
    const int base_col = rowbase[row];
    double acc = 0.0;
    for (int64_t base = b; base < e; base += 2 * KR_CHUNK) {           // same cells per lane, same order of the adds as kr_spmv
        const int64_t j0 = base + 4 * lane, j1 = j0 + KR_CHUNK;
        const bool second = base + KR_CHUNK < e;                        // wave-uniform
        const kr_u4 w0 = *reinterpret_cast<const kr_u4*>(pack + j0);
        kr_u4 w1 = w0;
        if (second) w1 = *reinterpret_cast<const kr_u4*>(pack + j1);
        KrCells<float> k0, k1;
        k0.c = kr_i4{base_col + (int)(w0.x & 0xFFFFu), base_col + (int)(w0.y & 0xFFFFu), base_col + (int)(w0.z & 0xFFFFu), base_col + (int)(w0.w & 0xFFFFu)};
        k0.f = kr_f4{(float)(w0.x >> 16), (float)(w0.y >> 16), (float)(w0.z >> 16), (float)(w0.w >> 16)};
        k1.c = kr_i4{base_col + (int)(w1.x & 0xFFFFu), base_col + (int)(w1.y & 0xFFFFu), base_col + (int)(w1.z & 0xFFFFu), base_col + (int)(w1.w & 0xFFFFu)};
        k1.f = kr_f4{(float)(w1.x >> 16), (float)(w1.y >> 16), (float)(w1.z >> 16), (float)(w1.w >> 16)};
        double x0[4], x1[4];
        kr_gather(k0, j0, e, in, x0);
        if (second) kr_gather(k1, j1, e, in, x1);
        acc = kr_fold(acc, k0, x0, j0, e);
        if (second) acc = kr_fold(acc, k1, x1, j1, e);
    }
    const double t = wave_tree_sum(acc);
    if (lane == 0) {
        if (MODE == 0) {
            out0[row] = t;
        } else if (MODE == 1) {
            const double v = a0[row] * t;
            out0[row] = v;
            out1[row] = 1.0 - v;
        } else {
            out0[row] = a0[row] * t + a1[row] * a2[row];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// n-sized vector kernels of the KR iteration; block = tile of 1024 elements, thread t owns t, t+256, t+512, t+768
// ---------------------------------------------------------------------------------------------------------------
#define KR_FOR_TILE(i) \
    for (int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x, k_ = 0; k_ < 4; ++k_, i += THREADS)

__global__ __launch_bounds__(THREADS) void kr_fill(double* a, double value, int64_t n) {
    KR_FOR_TILE(i) if (i < n) a[i] = value;
}

// sum a[i]*b[i]  (b == nullptr: sum a[i])
__global__ __launch_bounds__(THREADS) void kr_dot(const double* a, const double* b, int64_t n, double* partials) {
    double acc = 0.0;
    KR_FOR_TILE(i) if (i < n) acc = acc + (b ? a[i] * b[i] : a[i]);
    tile_reduce_store<0>(acc, partials);
}

// k == 1: Z = rk / v; p = Z; xp = x * p; partial rk.Z          (HiCKRy.py:180-184)
__global__ __launch_bounds__(THREADS) void kr_cg_first(const double* rk, const double* v, const double* x, double* Z, double* p,
                                                       double* xp, int64_t n, double* partials) {
    double acc = 0.0;
    KR_FOR_TILE(i) if (i < n) {
        const double z = rk[i] / v[i];
        Z[i] = z;
        p[i] = z;
        xp[i] = x[i] * z;
        acc = acc + rk[i] * z;
    }
    tile_reduce_store<0>(acc, partials);
}

// k > 1: p = Z + beta * p; xp = x * p                         (HiCKRy.py:186-187)
__global__ __launch_bounds__(THREADS) void kr_cg_next(const double* Z, const double* x, double* p, double* xp, double beta, int64_t n) {
    KR_FOR_TILE(i) if (i < n) {
        const double pn = Z[i] + beta * p[i];
        p[i] = pn;
        xp[i] = x[i] * pn;
    }
}

// ap = alpha * p; ynew = y + ap; partial min / max of ynew     (HiCKRy.py:195-198)
__global__ __launch_bounds__(THREADS) void kr_step_try(const double* p, const double* y, double* ap, double* ynew, double alpha,
                                                       int64_t n, double* pmin, double* pmax) {
    double lo = INFINITY, hi = -INFINITY;
    KR_FOR_TILE(i) if (i < n) {
        const double a = alpha * p[i];
        const double yn = y[i] + a;
        ap[i] = a;
        ynew[i] = yn;
        lo = nan_min(lo, yn);
        hi = nan_max(hi, yn);
    }
    tile_reduce_store<1>(lo, pmin);
    tile_reduce_store<2>(hi, pmax);
}

// y = ynew; rk -= alpha * w; Z = rk / v; partial rk.Z         (HiCKRy.py:214-219)
__global__ __launch_bounds__(THREADS) void kr_step_accept(const double* ynew, const double* w, const double* v, double* y, double* rk,
                                                          double* Z, double alpha, int64_t n, double* partials) {
    double acc = 0.0;
    KR_FOR_TILE(i) if (i < n) {
        y[i] = ynew[i];
        const double r = rk[i] - alpha * w[i];
        rk[i] = r;
        const double z = r / v[i];
        Z[i] = z;
        acc = acc + r * z;
    }
    tile_reduce_store<0>(acc, partials);
}

// boundary of the cone: gamma = min over the selected i of (bound - y) / ap;  which 0: ap < 0 (HiCKRy.py:203-204),
// which 1: ynew > bound (HiCKRy.py:209-210).  Also counts the selected elements (np.amin of nothing raises).
__global__ __launch_bounds__(THREADS) void kr_gamma(const double* y, const double* ap, const double* ynew, double bound, int which,
                                                    int64_t n, double* pmin, unsigned long long* n_selected) {
    double lo = INFINITY;
    unsigned int cnt = 0;
    KR_FOR_TILE(i) if (i < n) {
        const bool sel = which == 0 ? (ap[i] < 0.0) : (ynew[i] > bound);
        if (sel) {
            lo = nan_min(lo, (bound - y[i]) / ap[i]);
            ++cnt;
        }
    }
    tile_reduce_store<1>(lo, pmin);
    if (cnt) atomicAdd(n_selected, (unsigned long long)cnt);
}

// y += gamma * ap                                             (HiCKRy.py:205,211)
__global__ __launch_bounds__(THREADS) void kr_step_partial(double* y, const double* ap, double gamma, int64_t n) {
    KR_FOR_TILE(i) if (i < n) y[i] = y[i] + gamma * ap[i];
}

// x *= y                                                      (HiCKRy.py:221)
__global__ __launch_bounds__(THREADS) void kr_scale(double* x, const double* y, int64_t n) {
    KR_FOR_TILE(i) if (i < n) x[i] = x[i] * y[i];
}

// ---------------------------------------------------------------------------------------------------------------
// assembly
// ---------------------------------------------------------------------------------------------------------------
__device__ inline int64_t find_locus(const unsigned long long* keys, const int32_t* index, int64_t nl, unsigned long long key) {
    int64_t lo = 0, hi = nl;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return (lo < nl && keys[lo] == key) ? (int64_t)index[lo] : -1;
}

// keys[i] = x*n + y, keys[m+i] = y*n + x                      (HiCKRy.py:36-52: coo + its transpose)
__global__ __launch_bounds__(THREADS) void kr_lookup(int64_t m, const int32_t* chr1, const int32_t* mid1, const int32_t* chr2,
                                                     const int32_t* mid2, const unsigned long long* locus_keys,
                                                     const int32_t* locus_index, int64_t nl, int64_t n, unsigned long long* keys,
                                                     unsigned long long* first_missing) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = find_locus(locus_keys, locus_index, nl, ((unsigned long long)(unsigned int)chr1[i] << 32) | (unsigned int)mid1[i]);
        const int64_t y = find_locus(locus_keys, locus_index, nl, ((unsigned long long)(unsigned int)chr2[i] << 32) | (unsigned int)mid2[i]);
        if (x < 0 || y < 0) {
            atomicMin(first_missing, (unsigned long long)i);
            keys[i] = keys[m + i] = 0;
            continue;
        }
        keys[i] = (unsigned long long)(x * n + y);
        keys[m + i] = (unsigned long long)(y * n + x);
    }
}

using fhxscan::block_exclusive_scan;
using fhxscan::is_head;

// every run head adds its run one by one (stable sort => file order) and writes one CSR cell
__global__ __launch_bounds__(THREADS) void kr_emit_cells(const unsigned long long* keys, const unsigned int* perm, const double* z,
                                                         int64_t m, int64_t N, int64_t n, const unsigned long long* tile_offsets,
                                                         unsigned long long* cell_key, int32_t* col, double* val) {
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    unsigned int c = 0;
    bool head[SCAN_ITEMS];
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        head[k] = base + k < N && is_head(keys, base + k);
        c += head[k] ? 1u : 0u;
    }
    unsigned int total;
    unsigned long long pos = tile_offsets[blockIdx.x] + block_exclusive_scan(c, &total);
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (!head[k]) continue;
        const int64_t i = base + k;
        const unsigned long long key = keys[i];
        unsigned int src = perm[i];
        double s = z[src >= m ? src - m : src];
        for (int64_t j = i + 1; j < N && keys[j] == key; ++j) {
            src = perm[j];
            s = s + z[src >= m ? src - m : src];
        }
        cell_key[pos] = key;
        col[pos] = (int32_t)(key % (unsigned long long)n);
        val[pos] = s;
        ++pos;
    }
}

// indptr[r] = first cell whose key >= r*n
__global__ __launch_bounds__(THREADS) void kr_indptr(const unsigned long long* cell_key, int64_t nnz, int64_t n, int64_t* indptr) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long want = (unsigned long long)r * (unsigned long long)n;
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cell_key[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        indptr[r] = lo;
    }
}

// row / column removal (HiCKRy.py:94-101): newidx[old] = new index or -1
__global__ __launch_bounds__(THREADS) void kr_count_kept(int64_t n_old, const int64_t* indptr, const int32_t* col, const int32_t* newidx,
                                                         int64_t* counts_new) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_old) return;
    const int32_t nr = newidx[row];
    if (nr < 0) return;
    int c = 0;
    for (int64_t j = indptr[row] + lane; j < indptr[row + 1]; j += 64) c += newidx[col[j]] >= 0 ? 1 : 0;
    for (int s = 32; s >= 1; s >>= 1) c += __shfl_down(c, s, 64);
    if (lane == 0) counts_new[nr] = c;
}

__global__ __launch_bounds__(THREADS) void kr_compact(int64_t n_old, const int64_t* indptr, const int32_t* col, const double* val,
                                                      const int32_t* newidx, const int64_t* rptr, int32_t* rcol, double* rval) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_old) return;
    const int32_t nr = newidx[row];
    if (nr < 0) return;
    int64_t base = rptr[nr];
    const int64_t b = indptr[row], e = indptr[row + 1];
    for (int64_t j0 = b; j0 < e; j0 += 64) {
        const int64_t j = j0 + lane;
        const int32_t nc = j < e ? newidx[col[j]] : -1;
        const unsigned long long mask = __ballot(nc >= 0);
        if (nc >= 0) {
            const int64_t at = base + __popcll(mask & ((1ull << lane) - 1ull));
            rcol[at] = nc;
            rval[at] = val[j];
        }
        base += __popcll(mask);
    }
}

}  // namespace krd

// ===================================================================================================================
// host side
// ===================================================================================================================
struct fhx_kr {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    fhx_ctx* sorter = nullptr;
    // loci
    std::vector<unsigned long long> locus_keys;      // sorted distinct (chr<<32 | mid)
    std::vector<int32_t> locus_index;                // line number of the LAST occurrence (dict overwrite, HiCKRy.py:30)
    int64_t n_full = 0;
    // full matrix
    int64_t nnz_full = 0;
    int64_t* d_indptr = nullptr;
    int32_t* d_col = nullptr;
    double* d_val = nullptr;
    // reduced matrix (aliases the full one until fhx_kr_remove_sparse)
    bool reduced = false;
    int64_t n = 0, nnz = 0;
    int64_t* d_rptr = nullptr;
    int32_t* d_rcol = nullptr;
    double* d_rval = nullptr;
    float* d_val32 = nullptr;                          // binary32 copy of the matrix fhx_kr_balance streams (exact or absent)
    bool val32_reduced = false;
    unsigned int* d_pack = nullptr;                    // 4-byte cells (kr_pack16) of the same matrix, or absent
    int32_t* d_rowbase = nullptr;
    std::vector<int64_t> removed;
    std::vector<double> row_sums;
    // iteration state
    double* d_vec = nullptr;                          // 10 vectors of n doubles
    int64_t vec_cap = 0;
    double* d_part = nullptr;                         // 3 partial arrays
    int64_t part_cap = 0;
    double* d_scalars = nullptr;                      // 8 doubles
    unsigned long long* d_counter = nullptr;
    std::vector<double> x_host;
    bool balanced = false;
    fhx_kr_info info{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double spmv_seconds = 0;
    int64_t spmv_calls = 0;
};

namespace {

int kfail(fhx_kr* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define KR_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return kfail(kr, FHX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
void kfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

inline int tiles_of(int64_t n) { return (int)std::max<int64_t>(1, (n + krd::TILE - 1) / krd::TILE); }

const int64_t* mat_ptr(const fhx_kr* kr) { return kr->reduced ? kr->d_rptr : kr->d_indptr; }
const int32_t* mat_col(const fhx_kr* kr) { return kr->reduced ? kr->d_rcol : kr->d_col; }
const double* mat_val(const fhx_kr* kr) { return kr->reduced ? kr->d_rval : kr->d_val; }

template <int MODE>
void launch_spmv(fhx_kr* kr, int64_t n, const int64_t* ptr, const int32_t* col, const double* val, const double* in, double* out0,
                 double* out1, const double* a0, const double* a1, const double* a2) {
    const int64_t n_blocks = (n + 3) / 4;
    const int64_t per = (n_blocks + 7) / 8;
    const dim3 g((unsigned)std::max<int64_t>(1, per * 8)), t(krd::THREADS);
    if (kr->d_pack && val == (kr->val32_reduced ? kr->d_rval : kr->d_val))
        hipLaunchKernelGGL((krd::kr_spmv_packed<MODE>), g, t, 0, kr->stream, n, ptr, (const unsigned int*)kr->d_pack, (const int32_t*)kr->d_rowbase, in,
                           out0, out1, a0, a1, a2, n_blocks);
    else if (kr->d_val32 && val == (kr->val32_reduced ? kr->d_rval : kr->d_val))
        hipLaunchKernelGGL((krd::kr_spmv<MODE, float>), g, t, 0, kr->stream, n, ptr, col, (const float*)kr->d_val32, in, out0, out1, a0, a1,
                           a2, n_blocks);
    else
        hipLaunchKernelGGL((krd::kr_spmv<MODE, double>), g, t, 0, kr->stream, n, ptr, col, val, in, out0, out1, a0, a1, a2, n_blocks);
}

int ensure_vectors(fhx_kr* kr, int64_t n) {
    const int64_t cap = std::max<int64_t>(n, 1);
    if (cap > kr->vec_cap) {
        kfree(kr->d_vec);
        KR_HIP(hipMalloc(&kr->d_vec, (size_t)cap * 10 * sizeof(double)));
        kr->vec_cap = cap;
    }
    const int64_t t = tiles_of(cap);
    if (t > kr->part_cap) {
        kfree(kr->d_part);
        KR_HIP(hipMalloc(&kr->d_part, (size_t)t * 3 * sizeof(double)));
        kr->part_cap = t;
    }
    if (!kr->d_scalars) KR_HIP(hipMalloc(&kr->d_scalars, 8 * sizeof(double)));
    if (!kr->d_counter) KR_HIP(hipMalloc(&kr->d_counter, 4 * sizeof(unsigned long long)));
    return FHX_OK;
}

// finish the partial arrays `which` (bit mask over the 3 arrays; ops in `ops`) and bring the scalars to the host
int finish_scalars(fhx_kr* kr, int64_t n, int n_arrays, const int* ops, double* out) {
    const int64_t t = tiles_of(n);
    for (int a = 0; a < n_arrays; ++a) {
        const double* part = kr->d_part + (int64_t)a * kr->part_cap;
        if (ops[a] == 0) hipLaunchKernelGGL(krd::kr_finish<0>, dim3(1), dim3(krd::THREADS), 0, kr->stream, part, t, kr->d_scalars + a);
        else if (ops[a] == 1) hipLaunchKernelGGL(krd::kr_finish<1>, dim3(1), dim3(krd::THREADS), 0, kr->stream, part, t, kr->d_scalars + a);
        else hipLaunchKernelGGL(krd::kr_finish<2>, dim3(1), dim3(krd::THREADS), 0, kr->stream, part, t, kr->d_scalars + a);
    }
    KR_HIP(hipGetLastError());
    KR_HIP(hipMemcpyAsync(out, kr->d_scalars, n_arrays * sizeof(double), hipMemcpyDeviceToHost, kr->stream));
    KR_HIP(hipStreamSynchronize(kr->stream));
    return FHX_OK;
}

void free_matrix(fhx_kr* kr) {
    kfree(kr->d_indptr);
    kfree(kr->d_col);
    kfree(kr->d_val);
    kfree(kr->d_rptr);
    kfree(kr->d_rcol);
    kfree(kr->d_rval);
    kfree(kr->d_val32);
    kfree(kr->d_pack);
    kfree(kr->d_rowbase);
    kr->reduced = false;
    kr->balanced = false;
    kr->n = kr->nnz = kr->nnz_full = 0;
    kr->removed.clear();
    kr->row_sums.clear();
}

// numpy's pairwise np.sum of a contiguous double array (computeBiasVector, HiCKRy.py:106)
double numpy_pairwise(const double* a, int64_t n) {
    if (n < 8) {
        double r = -0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_pairwise(a, n2) + numpy_pairwise(a + n2, n - n2);
}

// np.sum of a contiguous double array: the reduction runs over buffers of 8192 elements (numpy's default buffer size), each
// summed pairwise, the buffer sums added from left to right (checked against numpy 2.2 up to 10^6 elements; a single pairwise
// pass over the whole array differs in the last bit for about half of the arrays beyond 8192 elements)
double numpy_sum(const double* a, int64_t n) {
    const int64_t B = 8192;
    if (n <= B) return numpy_pairwise(a, n);
    double acc = numpy_pairwise(a, B);
    for (int64_t i = B; i < n; i += B) acc += numpy_pairwise(a + i, std::min<int64_t>(B, n - i));
    return acc;
}

}  // namespace

extern "C" {

int fhx_kr_create(int device, fhx_kr** out) {
    if (!out) return FHX_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return FHX_ERR_NO_DEVICE;
    fhx_kr* kr = new fhx_kr();
    kr->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&kr->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&kr->ev0) != hipSuccess || hipEventCreate(&kr->ev1) != hipSuccess || fhx_create(device, &kr->sorter) != FHX_OK) {
        delete kr;
        return FHX_ERR_HIP;
    }
    *out = kr;
    return FHX_OK;
}

void fhx_kr_destroy(fhx_kr* kr) {
    if (!kr) return;
    (void)hipSetDevice(kr->device);
    if (kr->stream) (void)hipStreamSynchronize(kr->stream);
    free_matrix(kr);
    kfree(kr->d_vec);
    kfree(kr->d_part);
    kfree(kr->d_scalars);
    kfree(kr->d_counter);
    if (kr->ev0) (void)hipEventDestroy(kr->ev0);
    if (kr->ev1) (void)hipEventDestroy(kr->ev1);
    if (kr->sorter) fhx_destroy(kr->sorter);
    if (kr->stream) (void)hipStreamDestroy(kr->stream);
    delete kr;
}

const char* fhx_kr_last_error(const fhx_kr* kr) { return kr ? kr->err.c_str() : "null context"; }

int fhx_kr_load_loci(fhx_kr* kr, const int32_t* chr, const int32_t* mid, int64_t n) {
    if (!kr || n < 0 || (n > 0 && (!chr || !mid))) return FHX_ERR_ARG;
    if (n >= (1ll << 31)) return kfail(kr, FHX_ERR_UNSUPPORTED, "more than 2^31 loci");
    std::vector<std::pair<unsigned long long, int32_t>> kv((size_t)n);
    for (int64_t i = 0; i < n; ++i)
        kv[(size_t)i] = {((unsigned long long)(unsigned int)chr[i] << 32) | (unsigned int)mid[i], (int32_t)i};
    std::sort(kv.begin(), kv.end());                  // by key, then by line number
    kr->locus_keys.clear();
    kr->locus_index.clear();
    for (size_t i = 0; i < kv.size(); ++i) {
        if (i + 1 < kv.size() && kv[i + 1].first == kv[i].first) continue;      // keep the last line of a repeated locus
        kr->locus_keys.push_back(kv[i].first);
        kr->locus_index.push_back(kv[i].second);
    }
    kr->n_full = n;
    free_matrix(kr);
    return FHX_OK;
}

int fhx_kr_load_pairs(fhx_kr* kr, const int32_t* chr1, const int32_t* mid1, const int32_t* chr2, const int32_t* mid2,
                      const double* value, int64_t m, int64_t* first_unknown_row) {
    if (!kr || m < 0 || (m > 0 && (!chr1 || !mid1 || !chr2 || !mid2 || !value))) return FHX_ERR_ARG;
    if (first_unknown_row) *first_unknown_row = -1;
    if (2 * m >= (1ll << 32)) return kfail(kr, FHX_ERR_UNSUPPORTED, "more than 2^31 rows");
    const int64_t n = kr->n_full;
    if (n <= 0) return kfail(kr, FHX_ERR_ARG, "fhx_kr_load_loci must be called first");
    KR_HIP(hipSetDevice(kr->device));
    free_matrix(kr);
    const int64_t N = 2 * m;
    KR_HIP(hipMalloc(&kr->d_indptr, (size_t)(n + 1) * sizeof(int64_t)));
    if (m == 0) {
        KR_HIP(hipMemsetAsync(kr->d_indptr, 0, (size_t)(n + 1) * sizeof(int64_t), kr->stream));
        KR_HIP(hipMalloc(&kr->d_col, (size_t)(1 + krd::KR_PAD) * sizeof(int32_t)));
        KR_HIP(hipMalloc(&kr->d_val, (size_t)(1 + krd::KR_PAD) * sizeof(double)));
        KR_HIP(hipStreamSynchronize(kr->stream));
        kr->n = n;
        return FHX_OK;
    }
    int32_t *c1 = nullptr, *m1 = nullptr, *c2 = nullptr, *m2 = nullptr, *lidx = nullptr;
    double* z = nullptr;
    unsigned long long *keys = nullptr, *skeys = nullptr, *lkeys = nullptr, *tile_off = nullptr, *cell_key = nullptr;
    unsigned int *perm = nullptr, *tile_cnt = nullptr;
    int rc = FHX_OK;
    auto cleanup = [&]() {
        kfree(c1); kfree(m1); kfree(c2); kfree(m2); kfree(lidx); kfree(z); kfree(keys); kfree(skeys); kfree(lkeys);
        kfree(tile_off); kfree(cell_key); kfree(perm); kfree(tile_cnt);
    };
#define KR_TRY(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            cleanup();                                                                                       \
            return kfail(kr, FHX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                \
        }                                                                                                    \
    } while (0)
    const size_t nl = kr->locus_keys.size();
    KR_TRY(hipMalloc(&c1, (size_t)m * 4));
    KR_TRY(hipMalloc(&m1, (size_t)m * 4));
    KR_TRY(hipMalloc(&c2, (size_t)m * 4));
    KR_TRY(hipMalloc(&m2, (size_t)m * 4));
    KR_TRY(hipMalloc(&z, (size_t)m * 8));
    KR_TRY(hipMalloc(&lkeys, std::max<size_t>(1, nl) * 8));
    KR_TRY(hipMalloc(&lidx, std::max<size_t>(1, nl) * 4));
    KR_TRY(hipMalloc(&keys, (size_t)N * 8));
    KR_TRY(hipMemcpyAsync(c1, chr1, (size_t)m * 4, hipMemcpyHostToDevice, kr->stream));
    KR_TRY(hipMemcpyAsync(m1, mid1, (size_t)m * 4, hipMemcpyHostToDevice, kr->stream));
    KR_TRY(hipMemcpyAsync(c2, chr2, (size_t)m * 4, hipMemcpyHostToDevice, kr->stream));
    KR_TRY(hipMemcpyAsync(m2, mid2, (size_t)m * 4, hipMemcpyHostToDevice, kr->stream));
    KR_TRY(hipMemcpyAsync(z, value, (size_t)m * 8, hipMemcpyHostToDevice, kr->stream));
    KR_TRY(hipMemcpyAsync(lkeys, kr->locus_keys.data(), nl * 8, hipMemcpyHostToDevice, kr->stream));
    KR_TRY(hipMemcpyAsync(lidx, kr->locus_index.data(), nl * 4, hipMemcpyHostToDevice, kr->stream));
    if (!kr->d_counter) KR_TRY(hipMalloc(&kr->d_counter, 4 * sizeof(unsigned long long)));
    const unsigned long long none = ~0ull;
    KR_TRY(hipMemcpyAsync(kr->d_counter, &none, 8, hipMemcpyHostToDevice, kr->stream));
    const int lgrid = (int)std::min<int64_t>((m + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(krd::kr_lookup, dim3(lgrid), dim3(256), 0, kr->stream, m, c1, m1, c2, m2, lkeys, lidx, (int64_t)nl, n, keys,
                       kr->d_counter);
    KR_TRY(hipGetLastError());
    unsigned long long missing = none;
    KR_TRY(hipMemcpyAsync(&missing, kr->d_counter, 8, hipMemcpyDeviceToHost, kr->stream));
    KR_TRY(hipStreamSynchronize(kr->stream));
    kfree(c1); kfree(m1); kfree(c2); kfree(m2); kfree(lkeys); kfree(lidx);
    if (missing != none) {
        cleanup();
        if (first_unknown_row) *first_unknown_row = (int64_t)missing;
        return kfail(kr, FHX_ERR_REFERENCE_EXIT, "row " + std::to_string(missing) +
                     " names a locus that is not in the fragments file (the reference raises KeyError, HiCKRy.py:44-45)");
    }
    // stable sort of the 2m cell keys
    KR_TRY(hipMalloc(&skeys, (size_t)N * 8));
    KR_TRY(hipMalloc(&perm, (size_t)N * 4));
    rc = fhx_sort_u64(kr->sorter, keys, N, skeys, perm);
    if (rc != FHX_OK) {
        cleanup();
        return kfail(kr, rc, std::string("sort: ") + fhx_last_error(kr->sorter));
    }
    kfree(keys);
    // run heads -> cells
    const int64_t tiles = (N + krd::TILE - 1) / krd::TILE;
    KR_TRY(hipMalloc(&tile_cnt, (size_t)tiles * 4));
    KR_TRY(hipMalloc(&tile_off, (size_t)tiles * 8));
    hipLaunchKernelGGL(fhxscan::count_heads, dim3((unsigned)tiles), dim3(krd::THREADS), 0, kr->stream, skeys, N, tile_cnt);
    hipLaunchKernelGGL(fhxscan::scan_tiles, dim3(1), dim3(krd::THREADS), 0, kr->stream, tile_cnt, tiles, tile_off, kr->d_counter + 1);
    KR_TRY(hipGetLastError());
    unsigned long long nnz = 0;
    KR_TRY(hipMemcpyAsync(&nnz, kr->d_counter + 1, 8, hipMemcpyDeviceToHost, kr->stream));
    KR_TRY(hipStreamSynchronize(kr->stream));
    KR_TRY(hipMalloc(&cell_key, (size_t)nnz * 8));
    KR_TRY(hipMalloc(&kr->d_col, (size_t)(nnz + krd::KR_PAD) * 4));
    KR_TRY(hipMalloc(&kr->d_val, (size_t)(nnz + krd::KR_PAD) * 8));
    hipLaunchKernelGGL(krd::kr_emit_cells, dim3((unsigned)tiles), dim3(krd::THREADS), 0, kr->stream, skeys, perm, z, m, N, n, tile_off,
                       cell_key, kr->d_col, kr->d_val);
    hipLaunchKernelGGL(krd::kr_indptr, dim3((unsigned)std::min<int64_t>((n + 256) / 256, 4096)), dim3(256), 0, kr->stream, cell_key,
                       (int64_t)nnz, n, kr->d_indptr);
    KR_TRY(hipGetLastError());
    KR_TRY(hipStreamSynchronize(kr->stream));
    cleanup();
#undef KR_TRY
    kr->nnz_full = kr->nnz = (int64_t)nnz;
    kr->n = n;
    return FHX_OK;
}

int fhx_kr_shape(const fhx_kr* kr, int64_t* n_full, int64_t* nnz_full, int64_t* n_reduced, int64_t* nnz_reduced) {
    if (!kr) return FHX_ERR_ARG;
    if (n_full) *n_full = kr->n_full;
    if (nnz_full) *nnz_full = kr->nnz_full;
    if (n_reduced) *n_reduced = kr->n;
    if (nnz_reduced) *nnz_reduced = kr->nnz;
    return FHX_OK;
}

int fhx_kr_get_csr(fhx_kr* kr, int32_t which, int64_t* indptr, int32_t* col, double* val) {
    if (!kr || (which != 0 && which != 1)) return FHX_ERR_ARG;
    if (!kr->d_indptr) return kfail(kr, FHX_ERR_ARG, "no matrix loaded");
    if (which == 1 && !kr->reduced) return kfail(kr, FHX_ERR_ARG, "fhx_kr_remove_sparse has not run");
    KR_HIP(hipSetDevice(kr->device));
    const int64_t n = which ? kr->n : kr->n_full, nnz = which ? kr->nnz : kr->nnz_full;
    if (indptr) KR_HIP(hipMemcpyAsync(indptr, which ? kr->d_rptr : kr->d_indptr, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, kr->stream));
    if (col && nnz) KR_HIP(hipMemcpyAsync(col, which ? kr->d_rcol : kr->d_col, (size_t)nnz * 4, hipMemcpyDeviceToHost, kr->stream));
    if (val && nnz) KR_HIP(hipMemcpyAsync(val, which ? kr->d_rval : kr->d_val, (size_t)nnz * 8, hipMemcpyDeviceToHost, kr->stream));
    KR_HIP(hipStreamSynchronize(kr->stream));
    return FHX_OK;
}

int fhx_kr_spmv(fhx_kr* kr, int32_t which, const double* x, double* y, int32_t repeats, double* seconds_per_call) {
    if (!kr || !x || !y || (which != 0 && which != 1) || repeats < 1) return FHX_ERR_ARG;
    if (!kr->d_indptr) return kfail(kr, FHX_ERR_ARG, "no matrix loaded");
    if (which == 1 && !kr->reduced) return kfail(kr, FHX_ERR_ARG, "fhx_kr_remove_sparse has not run");
    KR_HIP(hipSetDevice(kr->device));
    const int64_t n = which ? kr->n : kr->n_full;
    int rc = ensure_vectors(kr, std::max(n, kr->vec_cap));
    if (rc != FHX_OK) return rc;
    double *dx = kr->d_vec, *dy = kr->d_vec + kr->vec_cap;
    KR_HIP(hipMemcpyAsync(dx, x, (size_t)n * 8, hipMemcpyHostToDevice, kr->stream));
    KR_HIP(hipEventRecord(kr->ev0, kr->stream));
    for (int r = 0; r < repeats; ++r)
        launch_spmv<0>(kr, n, which ? kr->d_rptr : kr->d_indptr, which ? kr->d_rcol : kr->d_col, which ? kr->d_rval : kr->d_val, dx, dy,
                       nullptr, nullptr, nullptr, nullptr);
    KR_HIP(hipEventRecord(kr->ev1, kr->stream));
    KR_HIP(hipGetLastError());
    KR_HIP(hipMemcpyAsync(y, dy, (size_t)n * 8, hipMemcpyDeviceToHost, kr->stream));
    KR_HIP(hipStreamSynchronize(kr->stream));
    float ms = 0;
    KR_HIP(hipEventElapsedTime(&ms, kr->ev0, kr->ev1));
    if (seconds_per_call) *seconds_per_call = 1e-3 * ms / repeats;
    kr->balanced = false;                             // the scratch vectors were reused
    return FHX_OK;
}

int fhx_kr_dot(fhx_kr* kr, const double* a, const double* b, int64_t n, double* out) {
    if (!kr || !a || !out || n < 0) return FHX_ERR_ARG;
    KR_HIP(hipSetDevice(kr->device));
    int rc = ensure_vectors(kr, std::max<int64_t>(n, kr->vec_cap));
    if (rc != FHX_OK) return rc;
    if (n == 0) {
        *out = 0.0;
        return FHX_OK;
    }
    double *da = kr->d_vec, *db = kr->d_vec + kr->vec_cap;
    KR_HIP(hipMemcpyAsync(da, a, (size_t)n * 8, hipMemcpyHostToDevice, kr->stream));
    if (b) KR_HIP(hipMemcpyAsync(db, b, (size_t)n * 8, hipMemcpyHostToDevice, kr->stream));
    hipLaunchKernelGGL(krd::kr_dot, dim3(tiles_of(n)), dim3(krd::THREADS), 0, kr->stream, da, b ? db : nullptr, n, kr->d_part);
    const int op = 0;
    kr->balanced = false;
    return finish_scalars(kr, n, 1, &op, out);
}

int fhx_kr_row_sums(fhx_kr* kr, double* out) {
    if (!kr) return FHX_ERR_ARG;
    if (!kr->d_indptr) return kfail(kr, FHX_ERR_ARG, "no matrix loaded");
    KR_HIP(hipSetDevice(kr->device));
    const int64_t n = kr->n_full;
    if (kr->row_sums.empty() && n > 0) {
        int rc = ensure_vectors(kr, std::max(n, kr->vec_cap));
        if (rc != FHX_OK) return rc;
        double *ones = kr->d_vec, *dy = kr->d_vec + kr->vec_cap;
        hipLaunchKernelGGL(krd::kr_fill, dim3(tiles_of(n)), dim3(krd::THREADS), 0, kr->stream, ones, 1.0, n);
        launch_spmv<0>(kr, n, kr->d_indptr, kr->d_col, kr->d_val, ones, dy, nullptr, nullptr, nullptr, nullptr);
        KR_HIP(hipGetLastError());
        kr->row_sums.resize((size_t)n);
        KR_HIP(hipMemcpyAsync(kr->row_sums.data(), dy, (size_t)n * 8, hipMemcpyDeviceToHost, kr->stream));
        KR_HIP(hipStreamSynchronize(kr->stream));
        kr->balanced = false;
    }
    if (out && n) std::memcpy(out, kr->row_sums.data(), (size_t)n * 8);
    return FHX_OK;
}

int fhx_kr_remove_sparse(fhx_kr* kr, double perc, int64_t* n_removed, double* val_to_remove, int64_t* rem_rows) {
    if (!kr) return FHX_ERR_ARG;
    if (!kr->d_indptr) return kfail(kr, FHX_ERR_ARG, "no matrix loaded");
    if (kr->reduced) return kfail(kr, FHX_ERR_ARG, "rows were already removed; load the pairs again");
    int rc = fhx_kr_row_sums(kr, nullptr);
    if (rc != FHX_OK) return rc;
    const int64_t n = kr->n_full;
    const int64_t rem = (int64_t)(perc * (double)n);              // int(perc * size), HiCKRy.py:83
    if (rem_rows) *rem_rows = rem;
    if (rem < 0 || rem >= n)
        return kfail(kr, FHX_ERR_REFERENCE_EXIT, "percentOfSparseToRemove selects row " + std::to_string(rem) + " of " +
                     std::to_string(n) + " (the reference raises IndexError, HiCKRy.py:86)");
    std::vector<double> sorted(kr->row_sums);
    std::nth_element(sorted.begin(), sorted.begin() + rem, sorted.end());
    const double val = sorted[(size_t)rem];
    std::vector<int32_t> newidx((size_t)n);
    kr->removed.clear();
    int32_t next = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (kr->row_sums[(size_t)i] <= val) {
            kr->removed.push_back(i);
            newidx[(size_t)i] = -1;
        } else {
            newidx[(size_t)i] = next++;
        }
    }
    if (n_removed) *n_removed = (int64_t)kr->removed.size();
    if (val_to_remove) *val_to_remove = val;
    const int64_t n2 = next;
    KR_HIP(hipSetDevice(kr->device));
    int32_t* d_new = nullptr;
    int64_t* d_cnt = nullptr;
    struct Scratch {                                   // freed on every return path
        int32_t*& a;
        int64_t*& b;
        ~Scratch() {
            kfree(a);
            kfree(b);
        }
    } scratch{d_new, d_cnt};
    KR_HIP(hipMalloc(&d_new, (size_t)n * 4));
    KR_HIP(hipMalloc(&d_cnt, (size_t)std::max<int64_t>(n2, 1) * 8));
    KR_HIP(hipMemcpyAsync(d_new, newidx.data(), (size_t)n * 4, hipMemcpyHostToDevice, kr->stream));
    const unsigned rows_grid = (unsigned)std::max<int64_t>(1, (n + 3) / 4);
    hipLaunchKernelGGL(krd::kr_count_kept, dim3(rows_grid), dim3(krd::THREADS), 0, kr->stream, n, kr->d_indptr, kr->d_col, d_new, d_cnt);
    std::vector<int64_t> rptr((size_t)n2 + 1, 0);
    KR_HIP(hipGetLastError());
    if (n2) KR_HIP(hipMemcpyAsync(rptr.data() + 1, d_cnt, (size_t)n2 * 8, hipMemcpyDeviceToHost, kr->stream));
    KR_HIP(hipStreamSynchronize(kr->stream));
    for (int64_t i = 0; i < n2; ++i) rptr[(size_t)i + 1] += rptr[(size_t)i];
    const int64_t nnz2 = rptr[(size_t)n2];
    KR_HIP(hipMalloc(&kr->d_rptr, (size_t)(n2 + 1) * 8));
    KR_HIP(hipMalloc(&kr->d_rcol, (size_t)(std::max<int64_t>(nnz2, 1) + krd::KR_PAD) * 4));
    KR_HIP(hipMalloc(&kr->d_rval, (size_t)(std::max<int64_t>(nnz2, 1) + krd::KR_PAD) * 8));
    KR_HIP(hipMemcpyAsync(kr->d_rptr, rptr.data(), (size_t)(n2 + 1) * 8, hipMemcpyHostToDevice, kr->stream));
    hipLaunchKernelGGL(krd::kr_compact, dim3(rows_grid), dim3(krd::THREADS), 0, kr->stream, n, kr->d_indptr, kr->d_col, kr->d_val, d_new,
                       kr->d_rptr, kr->d_rcol, kr->d_rval);
    KR_HIP(hipGetLastError());
    KR_HIP(hipStreamSynchronize(kr->stream));
    kr->reduced = true;
    kr->n = n2;
    kr->nnz = nnz2;
    kr->balanced = false;
    return FHX_OK;
}

int fhx_kr_get_removed(const fhx_kr* kr, int64_t* idx, int64_t capacity, int64_t* n_out) {
    if (!kr) return FHX_ERR_ARG;
    if (n_out) *n_out = (int64_t)kr->removed.size();
    if (idx) {
        if (capacity < (int64_t)kr->removed.size()) return FHX_ERR_ARG;
        std::copy(kr->removed.begin(), kr->removed.end(), idx);
    }
    return FHX_OK;
}

// knightRuizAlg (HiCKRy.py:139-243).  Scalars and control flow on the host, vectors on the device.
int fhx_kr_balance(fhx_kr* kr, double tol, fhx_kr_info* out) {
    if (!kr) return FHX_ERR_ARG;
    if (!kr->d_indptr) return kfail(kr, FHX_ERR_ARG, "no matrix loaded");
    KR_HIP(hipSetDevice(kr->device));
    const int64_t n = kr->n;
    int rc = ensure_vectors(kr, std::max<int64_t>(n, kr->vec_cap));
    if (rc != FHX_OK) return rc;
    const int64_t cap = kr->vec_cap;
    double *x = kr->d_vec, *v = x + cap, *rk = v + cap, *Z = rk + cap, *p = Z + cap, *w = p + cap, *y = w + cap, *ap = y + cap,
           *ynew = ap + cap, *xp = ynew + cap;
    double *part0 = kr->d_part, *part1 = kr->d_part + kr->part_cap;
    const int64_t* ptr = mat_ptr(kr);
    const int32_t* col = mat_col(kr);
    const double* val = mat_val(kr);
    // binary32 copy of the values when that is exact (integer contact counts): 8 B per cell instead of 12
    if (kr->nnz > 0 && (!kr->d_val32 || kr->val32_reduced != kr->reduced)) {
        kfree(kr->d_val32);
        KR_HIP(hipMalloc(&kr->d_val32, (size_t)(kr->nnz + krd::KR_PAD) * sizeof(float)));
        if (!kr->d_counter) KR_HIP(hipMalloc(&kr->d_counter, 4 * sizeof(unsigned long long)));
        unsigned int* flag = reinterpret_cast<unsigned int*>(kr->d_counter + 3);
        KR_HIP(hipMemsetAsync(flag, 0, 4, kr->stream));
        hipLaunchKernelGGL(krd::kr_to_f32, dim3((unsigned)std::min<int64_t>((kr->nnz + 255) / 256, 256 * 16)), dim3(krd::THREADS), 0,
                           kr->stream, val, kr->nnz, kr->d_val32, flag);
        unsigned int inexact = 0;
        KR_HIP(hipMemcpyAsync(&inexact, flag, 4, hipMemcpyDeviceToHost, kr->stream));
        KR_HIP(hipStreamSynchronize(kr->stream));
        kr->val32_reduced = kr->reduced;
        if (inexact) kfree(kr->d_val32);               // fractional or huge counts: stay with the doubles
        // 4-byte cells when the rows are narrow and the counts small (kr_pack16)
        kfree(kr->d_pack);
        kfree(kr->d_rowbase);
        if (kr->d_val32 && !std::getenv("FHX_KR_NO_PACK")) {
            KR_HIP(hipMalloc(&kr->d_pack, (size_t)(kr->nnz + krd::KR_PAD) * sizeof(unsigned int)));
            KR_HIP(hipMalloc(&kr->d_rowbase, (size_t)std::max<int64_t>(n, 1) * sizeof(int32_t)));
            KR_HIP(hipMemsetAsync(kr->d_pack + kr->nnz, 0, (size_t)krd::KR_PAD * sizeof(unsigned int), kr->stream));
            KR_HIP(hipMemsetAsync(flag, 0, 4, kr->stream));
            hipLaunchKernelGGL(krd::kr_pack16, dim3((unsigned)((n + 3) / 4)), dim3(krd::THREADS), 0, kr->stream, n, ptr, col, val, kr->d_pack,
                               kr->d_rowbase, flag);
            unsigned int misfit = 0;
            KR_HIP(hipMemcpyAsync(&misfit, flag, 4, hipMemcpyDeviceToHost, kr->stream));
            KR_HIP(hipStreamSynchronize(kr->stream));
            if (misfit) {                              // a row wider than 65 535 columns or a count of 65 536 or more
                kfree(kr->d_pack);
                kfree(kr->d_rowbase);
            }
        }
    }
    const dim3 grid(tiles_of(n)), block(krd::THREADS);
    const int OPS_SUM[1] = {0}, OPS_MINMAX[2] = {1, 2}, OPS_MIN[1] = {1};
    kr->spmv_seconds = 0;
    kr->spmv_calls = 0;
    bool timing_pending = false;
    auto timed_spmv_begin = [&]() { (void)hipEventRecord(kr->ev0, kr->stream); };
    auto timed_spmv_end = [&]() {
        (void)hipEventRecord(kr->ev1, kr->stream);
        timing_pending = true;
    };
    auto collect_timing = [&]() {                       // after a stream sync
        if (!timing_pending) return;
        float ms = 0;
        if (hipEventElapsedTime(&ms, kr->ev0, kr->ev1) == hipSuccess) {
            kr->spmv_seconds += 1e-3 * ms;
            kr->spmv_calls += 1;
        }
        timing_pending = false;
    };
    fhx_kr_info info{};
    info.n = n;
    info.nnz = kr->nnz;
    if (n == 0) {
        kr->x_host.clear();
        kr->balanced = true;
        kr->info = info;
        if (out) *out = info;
        return FHX_OK;
    }

    const double Delta = 3, delta = 0.1, g = 0.9;
    const double etamax = 0.1;
    double eta = 0.1;
    const double stop_tol = tol * 0.5;
    const double rt = std::pow(tol, 2.0);
    double s[3];
    // x = e; v = x * A x; rk = 1 - v; rho = rk.rk
    hipLaunchKernelGGL(krd::kr_fill, grid, block, 0, kr->stream, x, 1.0, n);
    timed_spmv_begin();
    launch_spmv<1>(kr, n, ptr, col, val, x, v, rk, x, nullptr, nullptr);
    timed_spmv_end();
    hipLaunchKernelGGL(krd::kr_dot, grid, block, 0, kr->stream, rk, rk, n, part0);
    rc = finish_scalars(kr, n, 1, OPS_SUM, s);
    if (rc != FHX_OK) return rc;
    collect_timing();
    double rho_km1 = s[0], rho_km2 = rho_km1;
    double rout = rho_km1, rold = rho_km1;
    int i = 0, k = 0;
    int64_t mvp = 1;
    while (rout > rt) {
        ++i;
        if (i > 30) break;
        k = 0;
        hipLaunchKernelGGL(krd::kr_fill, grid, block, 0, kr->stream, y, 1.0, n);
        const double innertol = std::max(std::pow(eta, 2.0) * rout, rt);
        while (rho_km1 > innertol) {
            ++k;
            if (k == 1) {
                hipLaunchKernelGGL(krd::kr_cg_first, grid, block, 0, kr->stream, rk, v, x, Z, p, xp, n, part0);
                rc = finish_scalars(kr, n, 1, OPS_SUM, s);
                if (rc != FHX_OK) return rc;
                rho_km1 = s[0];
            } else {
                const double beta = rho_km1 / rho_km2;
                hipLaunchKernelGGL(krd::kr_cg_next, grid, block, 0, kr->stream, Z, x, p, xp, beta, n);
            }
            if (k > 10) break;
            timed_spmv_begin();
            launch_spmv<2>(kr, n, ptr, col, val, xp, w, nullptr, x, v, p);
            timed_spmv_end();
            ++mvp;
            hipLaunchKernelGGL(krd::kr_dot, grid, block, 0, kr->stream, p, w, n, part0);
            rc = finish_scalars(kr, n, 1, OPS_SUM, s);
            if (rc != FHX_OK) return rc;
            collect_timing();
            const double alpha = rho_km1 / s[0];
            hipLaunchKernelGGL(krd::kr_step_try, grid, block, 0, kr->stream, p, y, ap, ynew, alpha, n, part0, part1);
            rc = finish_scalars(kr, n, 2, OPS_MINMAX, s);
            if (rc != FHX_OK) return rc;
            const double ymin = s[0], ymax = s[1];
            int boundary = -1;
            double bound = 0;
            if (ymin <= delta) {
                boundary = 0;
                bound = delta;
            } else if (ymax >= Delta) {
                boundary = 1;
                bound = Delta;
            }
            if (boundary >= 0) {
                KR_HIP(hipMemsetAsync(kr->d_counter + 2, 0, 8, kr->stream));
                hipLaunchKernelGGL(krd::kr_gamma, grid, block, 0, kr->stream, y, ap, ynew, bound, boundary, n, part0, kr->d_counter + 2);
                rc = finish_scalars(kr, n, 1, OPS_MIN, s);
                if (rc != FHX_OK) return rc;
                unsigned long long selected = 0;
                KR_HIP(hipMemcpy(&selected, kr->d_counter + 2, 8, hipMemcpyDeviceToHost));
                if (selected == 0)
                    return kfail(kr, FHX_ERR_REFERENCE_EXIT, "np.amin of an empty selection at the cone boundary (the reference raises "
                                                              "ValueError, HiCKRy.py:204/210)");
                hipLaunchKernelGGL(krd::kr_step_partial, grid, block, 0, kr->stream, y, ap, s[0], n);
                info.boundary_steps += 1;
                break;
            }
            hipLaunchKernelGGL(krd::kr_step_accept, grid, block, 0, kr->stream, ynew, w, v, y, rk, Z, alpha, n, part0);
            rc = finish_scalars(kr, n, 1, OPS_SUM, s);
            if (rc != FHX_OK) return rc;
            rho_km2 = rho_km1;
            rho_km1 = s[0];
        }
        hipLaunchKernelGGL(krd::kr_scale, grid, block, 0, kr->stream, x, y, n);
        timed_spmv_begin();
        launch_spmv<1>(kr, n, ptr, col, val, x, v, rk, x, nullptr, nullptr);
        timed_spmv_end();
        ++mvp;
        hipLaunchKernelGGL(krd::kr_dot, grid, block, 0, kr->stream, rk, rk, n, part0);
        rc = finish_scalars(kr, n, 1, OPS_SUM, s);
        if (rc != FHX_OK) return rc;
        collect_timing();
        rho_km1 = s[0];
        rout = rho_km1;
        const double rat = rout / rold;
        rold = rout;
        const double res_norm = std::pow(rout, 0.5);
        const double eta_o = eta;
        eta = g * rat;
        if (g * std::pow(eta_o, 2.0) > 0.1) eta = std::max(eta, g * std::pow(eta_o, 2.0));
        eta = std::max(std::min(eta, etamax), stop_tol / res_norm);
    }
    KR_HIP(hipGetLastError());
    kr->x_host.resize((size_t)n);
    KR_HIP(hipMemcpyAsync(kr->x_host.data(), x, (size_t)n * 8, hipMemcpyDeviceToHost, kr->stream));
    KR_HIP(hipStreamSynchronize(kr->stream));
    info.outer_iterations = i;
    info.inner_iterations = k;
    info.matvecs = mvp;
    info.residual = rout;
    info.spmv_seconds = kr->spmv_seconds;
    info.spmv_timed = kr->spmv_calls;
    info.value_bytes = kr->d_pack ? 2 : (kr->d_val32 ? 4 : 8);
    kr->info = info;
    kr->balanced = true;
    if (out) *out = info;
    return FHX_OK;
}

int fhx_kr_get_x(const fhx_kr* kr, double* x) {
    if (!kr || !x) return FHX_ERR_ARG;
    if (!kr->balanced) return FHX_ERR_ARG;
    std::copy(kr->x_host.begin(), kr->x_host.end(), x);
    return FHX_OK;
}

// computeBiasVector + addZeroBiases (HiCKRy.py:103-115)
int fhx_kr_bias(fhx_kr* kr, double* bias) {
    if (!kr || !bias) return FHX_ERR_ARG;
    if (!kr->balanced) return kfail(kr, FHX_ERR_ARG, "fhx_kr_balance has not run");
    const int64_t n2 = kr->n;
    std::vector<double> inv((size_t)n2);
    for (int64_t i = 0; i < n2; ++i) inv[(size_t)i] = 1.0 / kr->x_host[(size_t)i];
    const double sums = numpy_sum(inv.data(), n2);
    const double avg = (1.0 * sums) / (double)n2;
    size_t r = 0;
    int64_t j = 0;
    for (int64_t i = 0; i < kr->n_full; ++i) {
        if (kr->reduced && r < kr->removed.size() && kr->removed[r] == i) {
            bias[i] = -1.0;
            ++r;
        } else {
            bias[i] = inv[(size_t)j++] / avg;
        }
    }
    return FHX_OK;
}

}  // extern "C"
