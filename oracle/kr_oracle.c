/* TEST INFRASTRUCTURE - CPU oracle helpers for the Knight-Ruiz balancing path (HiCKRy).
 *
 * The reference (fithic/utils/HiCKRy.py:139-243) does its sums through scipy's csr_matvec (sequential per row) and
 * BLAS ddot (order chosen by the BLAS build), so its last bits are machine dependent.  The engine fixes ONE summation
 * order per operation (the one its wave64 kernels use) and this file restates exactly that order in plain C, so that
 * GPU == oracle bit for bit while oracle vs reference is pinned by the golden vectors within a stated tolerance.
 *
 *   fho_kr_segsum : duplicates of one (row, col) key are added one by one in file order   (HiCKRy.py:50-51, coo->csr)
 *   fho_kr_spmv   : row sum = 64 sequential partials (chunks of 256 cells, partial l takes cells 4l..4l+3 of each chunk),
 *                   then a binary tree over the partials
 *   fho_kr_dot/sum: tiles of 1024; 4 sequential terms per thread, tree over each 64-lane wave, 4 waves left to right,
 *                   tile partials left to right
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.                              */
#include <stdint.h>
#include <stddef.h>

int64_t fho_kr_segsum(int64_t m, const int64_t* keys_sorted, const double* vals_sorted, int64_t* out_keys, double* out_vals) {
    int64_t n = 0;
    for (int64_t i = 0; i < m;) {
        int64_t j = i + 1;
        double s = vals_sorted[i];
        while (j < m && keys_sorted[j] == keys_sorted[i]) s = s + vals_sorted[j++];
        out_keys[n] = keys_sorted[i];
        out_vals[n++] = s;
        i = j;
    }
    return n;
}

static double wave_tree(double* v) {          /* v[64] is clobbered */
    for (int s = 32; s >= 1; s >>= 1)
        for (int l = 0; l < s; ++l) v[l] = v[l] + v[l + s];
    return v[0];
}

void fho_kr_spmv(int64_t n, const int64_t* indptr, const int32_t* indices, const double* data, const double* x, double* y) {
    for (int64_t i = 0; i < n; ++i) {
        double lane[64];
        for (int l = 0; l < 64; ++l) lane[l] = 0.0;
        const int64_t b = indptr[i], e = indptr[i + 1];
        for (int64_t j = b; j < e; ++j) {
            const double p = data[j] * x[indices[j]];
            const int l = (int)(((j - b) & 255) >> 2);     /* chunks of 256 cells, lane l owns cells 4l..4l+3 of each */
            lane[l] = lane[l] + p;
        }
        y[i] = wave_tree(lane);
    }
}

/* sum_i a[i]*b[i] (b == NULL: sum_i a[i]) in the tile order described above */
double fho_kr_dot(int64_t n, const double* a, const double* b) {
    double total = 0.0;
    for (int64_t t0 = 0; t0 < n; t0 += 1024) {
        double th[256];
        for (int t = 0; t < 256; ++t) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) {
                const int64_t i = t0 + t + 256 * k;
                if (i < n) acc = acc + (b ? a[i] * b[i] : a[i]);
            }
            th[t] = acc;
        }
        double tile = 0.0;
        for (int w = 0; w < 4; ++w) {
            const double r = wave_tree(th + 64 * w);
            tile = (w == 0) ? r : tile + r;
        }
        total = (t0 == 0) ? tile : total + tile;
    }
    return total;
}
