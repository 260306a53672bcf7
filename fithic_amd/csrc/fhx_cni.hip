// fhx_cni.hip - merging of nearby significant contacts on MI355X (gfx950): connected-component labelling of the significant
// bin pairs and greedy pick of representatives (reference: fithic/utils/CombineNearbyInteraction.py; SURVEY.md 8f rank 4,
// the step after Fit-Hi-C).
//
// The reference compares every pair of nodes of a chromosome in Python (O(n^2), lines 324-340) and walks sets and heaps.
// Here the nodes are the distinct cells of a bin lattice, so every relation is a lookup in one sorted key array:
//
//   nodes      [301-312]  key = chr<<48 | lo<<24 | hi (bin indices on the lattice); stable radix sort; the first row of a
//                         cell keeps its values (dict.setdefault)                          cn_keys, sort, cn_emit_nodes
//   components [324-354]  8 / 4 neighbours by binary search, lock-free union-find (CAS hooks the larger root under the
//                         smaller, so the root is the component's smallest node)           cn_union, cn_flatten
//   statistics [367-402]  size, first row, bounding box, sum of CC (integer atomics), number of the chromosome's cells
//                         inside the box (one range count per box row)                     cn_stats, cn_box_count
//   order      [470-480]  (component, q or -q, -CC, lo, hi): three stable radix sorts (CC, q, root) on top of the key order
//   pick       [497-516]  greedy: a node is picked iff no earlier node of its component within -n bins in both coordinates
//                         is picked.  Run as rounds over all undecided nodes: decided as soon as every earlier node in
//                         its window is decided (the greedy result is unique, so the rounds reproduce it)   cn_pick_round
//
// Everything is integer / index work: results are bit-exact by construction.  The output order (chromosome, component size,
// first row, rank) is two more stable sorts; the host only moves the finished records and formats nothing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/fithic_mi355x.h"
#include "fhx_scan.hpp"

namespace cnd {

constexpr int THREADS = 256;
constexpr int BIN_BITS = 24;
constexpr unsigned long long BIN_MASK = (1ull << BIN_BITS) - 1;

__host__ __device__ inline unsigned long long make_key(unsigned int chr, unsigned int lo, unsigned int hi) {
    return ((unsigned long long)chr << (2 * BIN_BITS)) | ((unsigned long long)lo << BIN_BITS) | hi;
}

// rows -> cell keys: bin index k = (N - r) / res, orientation (min, max)                      (:301-309)
__global__ __launch_bounds__(THREADS) void cn_keys(int64_t rows, const int32_t* chr, const int64_t* n1, const int64_t* n2, int64_t res,
                                                   int64_t r, unsigned long long* keys, unsigned int* bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = n1[i] - r, b = n2[i] - r;
        if (a < 0 || b < 0 || a % res != 0 || b % res != 0 || a / res > (int64_t)BIN_MASK || b / res > (int64_t)BIN_MASK ||
            chr[i] < 0 || chr[i] > 0xffff) {
            atomicAdd(bad, 1u);
            keys[i] = 0;
            continue;
        }
        const unsigned int k1 = (unsigned int)(a / res), k2 = (unsigned int)(b / res);
        keys[i] = make_key((unsigned int)chr[i], min(k1, k2), max(k1, k2));
    }
}

// one node per run of equal keys; the run's first element is the first row of that cell (stable sort)   (:312)
__global__ __launch_bounds__(fhxscan::THREADS) void cn_emit_nodes(const unsigned long long* keys, const unsigned int* perm, int64_t N,
                                                                  const unsigned long long* tile_offsets, unsigned long long* node_key,
                                                                  unsigned int* node_row) {
    const int64_t base = (int64_t)blockIdx.x * fhxscan::TILE + (int64_t)threadIdx.x * fhxscan::SCAN_ITEMS;
    unsigned int c = 0;
    bool head[fhxscan::SCAN_ITEMS];
    for (int k = 0; k < fhxscan::SCAN_ITEMS; ++k) {
        head[k] = base + k < N && fhxscan::is_head(keys, base + k);
        c += head[k] ? 1u : 0u;
    }
    unsigned int total;
    unsigned long long pos = tile_offsets[blockIdx.x] + fhxscan::block_exclusive_scan(c, &total);
    for (int k = 0; k < fhxscan::SCAN_ITEMS; ++k) {
        if (!head[k]) continue;
        node_key[pos] = keys[base + k];
        node_row[pos] = perm[base + k];
        ++pos;
    }
}

__device__ inline int64_t find_node(const unsigned long long* node_key, int64_t n, unsigned long long key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (node_key[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && node_key[lo] == key) ? lo : -1;
}

// parent[] is read and written by many workgroups in one launch: agent-scope relaxed atomics keep the accesses out of the
// non-coherent per-CU cache (a stale "I am a root" would make the CAS below fail forever)
__device__ inline unsigned int uf_load(unsigned int* parent, unsigned int i) {
    return __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline unsigned int uf_find(unsigned int* parent, unsigned int a) {
    unsigned int p = uf_load(parent, a);
    while (p != a) {
        const unsigned int g = uf_load(parent, p);
        if (g != p) __hip_atomic_store(parent + a, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving: g is an ancestor
        a = p;
        p = g;
    }
    return a;
}

__device__ inline void uf_union(unsigned int* parent, unsigned int a, unsigned int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const unsigned int t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&parent[a], a, b) == a) return;   // only roots are hooked, always under the smaller index
    }
}

// edges of the 8 / 4 neighbourhood (:324-340); each edge is taken from its larger end
__global__ __launch_bounds__(THREADS) void cn_union(const unsigned long long* node_key, int64_t n, int conn, unsigned int* parent) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long key = node_key[u];
        const int64_t lo = (int64_t)((key >> BIN_BITS) & BIN_MASK), hi = (int64_t)(key & BIN_MASK);
        const unsigned int chr = (unsigned int)(key >> (2 * BIN_BITS));
        for (int d0 = -1; d0 <= 1; ++d0)
            for (int d1 = -1; d1 <= 1; ++d1) {
                if ((d0 == 0 && d1 == 0) || (conn == 4 && d0 != 0 && d1 != 0)) continue;
                const int64_t a = lo + d0, b = hi + d1;
                if (a < 0 || b < 0 || a > b || b > (int64_t)BIN_MASK) continue;      // stored keys are (min, max)
                const unsigned long long other = make_key(chr, (unsigned int)a, (unsigned int)b);
                if (other >= key) continue;
                const int64_t v = find_node(node_key, n, other);
                if (v >= 0) uf_union(parent, (unsigned int)u, (unsigned int)v);
            }
    }
}

// root[u] into a second array with a read-only walk: a walk that also compresses paths could overwrite, late, the final
// value another thread has just stored for the same node
__global__ __launch_bounds__(THREADS) void cn_flatten(const unsigned int* parent, int64_t n, unsigned int* root) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x) {
        unsigned int a = (unsigned int)u, p = parent[a];
        while (p != a) {
            a = p;
            p = parent[a];
        }
        root[u] = a;
    }
}

struct CompStats {            // indexed by root node
    unsigned int size;
    unsigned int first_row;
    unsigned int min_lo, max_lo, min_hi, max_hi;
    long long sum_cc;
    unsigned long long have;
};

__global__ __launch_bounds__(THREADS) void cn_init_stats(CompStats* st, unsigned int* parent, int64_t n) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x) {
        st[u] = CompStats{0u, 0xffffffffu, 0xffffffffu, 0u, 0xffffffffu, 0u, 0ll, 0ull};
        parent[u] = (unsigned int)u;
    }
}

// size, first row, bounding box, sum of CC per component (:367-390); node values come from the cell's first row
__global__ __launch_bounds__(THREADS) void cn_stats(const unsigned long long* node_key, const unsigned int* node_row, const unsigned int* root,
                                                    const int64_t* cc, int64_t n, CompStats* st) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long key = node_key[u];
        const unsigned int lo = (unsigned int)((key >> BIN_BITS) & BIN_MASK), hi = (unsigned int)(key & BIN_MASK);
        CompStats* s = st + root[u];
        atomicAdd(&s->size, 1u);
        atomicMin(&s->first_row, node_row[u]);
        atomicMin(&s->min_lo, lo);
        atomicMax(&s->max_lo, lo);
        atomicMin(&s->min_hi, hi);
        atomicMax(&s->max_hi, hi);
        atomicAdd((unsigned long long*)&s->sum_cc, (unsigned long long)cc[node_row[u]]);
    }
}

// cells of the chromosome inside the component's bounding box (:392-398): one range count per box row
__global__ __launch_bounds__(THREADS) void cn_box_count(const unsigned long long* node_key, const unsigned int* root, int64_t n, CompStats* st) {
    const int lane = threadIdx.x & 63;
    for (int64_t u = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; u < n; u += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        if (root[u] != (unsigned int)u) continue;         // one wave per component
        const CompStats s = st[u];
        const unsigned int chr = (unsigned int)(node_key[u] >> (2 * BIN_BITS));
        unsigned long long cnt = 0;
        for (int64_t a = (int64_t)s.min_lo + lane; a <= (int64_t)s.max_lo; a += 64) {
            const unsigned long long k0 = make_key(chr, (unsigned int)a, s.min_hi), k1 = make_key(chr, (unsigned int)a, s.max_hi);
            int64_t lo = 0, hi = n;
            while (lo < hi) {                             // first key >= k0
                const int64_t mid = (lo + hi) >> 1;
                if (node_key[mid] < k0) lo = mid + 1;
                else hi = mid;
            }
            int64_t lo2 = lo, hi2 = n;
            while (lo2 < hi2) {                           // first key > k1
                const int64_t mid = (lo2 + hi2) >> 1;
                if (node_key[mid] <= k1) lo2 = mid + 1;
                else hi2 = mid;
            }
            cnt += (unsigned long long)(lo2 - lo);
        }
        for (int s2 = 32; s2 >= 1; s2 >>= 1) cnt += __shfl_down(cnt, s2, 64);
        if (lane == 0) st[u].have = cnt;
    }
}

// sort keys of the ranking (:476-480): list comparison of [q or -q, -CC, lo, hi]
__device__ inline unsigned long long ordered_bits(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | (1ull << 63));
}

__global__ __launch_bounds__(THREADS) void cn_key_cc(const unsigned int* node_row, const int64_t* cc, int64_t n, unsigned long long* out) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x)
        out[u] = (unsigned long long)(-cc[node_row[u]]) ^ (1ull << 63);
}

__global__ __launch_bounds__(THREADS) void cn_key_q(const unsigned int* order_in, const unsigned int* node_row, const double* q, int negate,
                                                    int64_t n, unsigned long long* out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const double v = q[node_row[order_in[t]]];
        out[t] = ordered_bits(negate ? -v : v);
    }
}

__global__ __launch_bounds__(THREADS) void cn_key_root(const unsigned int* order_in, const unsigned int* root, int64_t n, unsigned long long* out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) out[t] = root[order_in[t]];
}

// order_out[t] = order_in[perm[t]]
__global__ __launch_bounds__(THREADS) void cn_compose(const unsigned int* order_in, const unsigned int* perm, int64_t n, unsigned int* order_out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        order_out[t] = order_in ? order_in[perm[t]] : perm[t];
}

__global__ __launch_bounds__(THREADS) void cn_positions(const unsigned int* order, int64_t n, unsigned int* pos) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) pos[order[t]] = (unsigned int)t;
}

// state 0 undecided, 1 picked, 2 dropped.  Nodes past their component's candidate limit start as dropped (:508-509).
__global__ __launch_bounds__(THREADS) void cn_pick_init(const unsigned int* root, const unsigned int* pos, const unsigned int* seg_start,
                                                        const unsigned int* limit, int64_t n, unsigned char* state) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int r = root[u];
        state[u] = (pos[u] - seg_start[r]) < limit[r] ? 0 : 2;
    }
}

__global__ __launch_bounds__(THREADS) void cn_pick_round(const unsigned long long* node_key, const unsigned int* root, const unsigned int* pos,
                                                         int64_t n, int neigh, unsigned char* state, unsigned long long* undecided) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x) {
        if (state[u] != 0) continue;
        const unsigned long long key = node_key[u];
        const int64_t lo = (int64_t)((key >> BIN_BITS) & BIN_MASK), hi = (int64_t)(key & BIN_MASK);
        const unsigned int chr = (unsigned int)(key >> (2 * BIN_BITS));
        const unsigned int r = root[u], my = pos[u];
        bool dropped = false, blocked = false;
        for (int64_t a = max((int64_t)0, lo - neigh); a <= lo + neigh && !dropped; ++a) {
            // all cells of box row a in [hi-neigh, hi+neigh]: one search, then a short forward walk
            const int64_t b0 = max(a, max((int64_t)0, hi - neigh)), b1 = min((int64_t)BIN_MASK, hi + neigh);
            if (a > (int64_t)BIN_MASK || b0 > b1) continue;
            const unsigned long long k0 = make_key(chr, (unsigned int)a, (unsigned int)b0), k1 = make_key(chr, (unsigned int)a, (unsigned int)b1);
            int64_t l = 0, h = n;
            while (l < h) {
                const int64_t mid = (l + h) >> 1;
                if (node_key[mid] < k0) l = mid + 1;
                else h = mid;
            }
            for (int64_t v = l; v < n && node_key[v] <= k1; ++v) {
                if (v == u || root[v] != r || pos[v] > my) continue;
                const unsigned char sv = state[v];
                if (sv == 1) {
                    dropped = true;
                    break;
                }
                if (sv == 0) blocked = true;
            }
        }
        if (dropped) state[u] = 2;
        else if (!blocked) state[u] = 1;
        else atomicAdd(undecided, 1ull);
    }
}

// segments of the final order (grouped by root): where each component starts; also counts the components
__global__ __launch_bounds__(THREADS) void cn_seg_starts(const unsigned int* order, const unsigned int* root, int64_t n, unsigned int* seg_start,
                                                         unsigned long long* n_components) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int r = root[order[t]];
        if (t == 0 || root[order[t - 1]] != r) {
            seg_start[r] = (unsigned int)t;
            atomicAdd(n_components, 1ull);
        }
    }
}

// per component: how many leading candidates the greedy loop looks at.  -p 100: all.  0 < -p < 100: custom_percent (:36-52)
// of the component's q list gives a bound; the loop stops at the first element failing it (:508-509)
__global__ __launch_bounds__(THREADS) void cn_limits(const unsigned int* root, const unsigned int* order, const unsigned int* node_row,
                                                     const double* q, const CompStats* st, const unsigned int* seg_start, int64_t n,
                                                     int top_percent, int sort_order, unsigned int* limit, unsigned long long* largest) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x) {
        if (root[u] != (unsigned int)u) continue;
        const unsigned int L = st[u].size, s0 = seg_start[u];
        atomicMax(largest, (unsigned long long)L);
        unsigned int lim = L;
        if (top_percent < 100) {
            const long long index = ((long long)L * top_percent) / 100;
            const unsigned int at = index <= 1 ? L - 1 : (unsigned int)index;
            const double bound = q[node_row[order[s0 + at]]];
            if (sort_order == 0) {                        // ascending q: first element with q > bound
                unsigned int lo = 0, hi = L;
                while (lo < hi) {
                    const unsigned int mid = (lo + hi) / 2;
                    if (q[node_row[order[s0 + mid]]] > bound) hi = mid;
                    else lo = mid + 1;
                }
                lim = lo;
            } else {                                      // keys are -q: the first one decides, later ones are larger
                lim = (-q[node_row[order[s0]]] < bound) ? 0u : L;
            }
        }
        limit[u] = lim;
    }
}

// output order of the components (:204-212, 354): chromosome, size (largest first), first row.  Two stable sorts.
__global__ __launch_bounds__(THREADS) void cn_key_first_row(const unsigned int* root, const CompStats* st, int64_t n, unsigned long long* out) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x)
        out[u] = root[u] == (unsigned int)u ? (unsigned long long)st[u].first_row : ~0ull;
}

__global__ __launch_bounds__(THREADS) void cn_key_chr_size(const unsigned int* nodes_in, const unsigned long long* node_key, const unsigned int* root,
                                                           const CompStats* st, int64_t n, unsigned long long* out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int u = nodes_in[t];
        out[t] = root[u] == u ? ((node_key[u] >> (2 * BIN_BITS)) << 32) | (unsigned long long)(0xffffffffu - st[u].size) : ~0ull;
    }
}

__global__ __launch_bounds__(THREADS) void cn_comp_ranks(const unsigned int* comps_sorted, int64_t n_components, unsigned int* comp_rank) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_components; t += (int64_t)gridDim.x * blockDim.x)
        comp_rank[comps_sorted[t]] = (unsigned int)t;
}

// picked nodes sort by (component rank, position in the component's ranking); the rest goes to the end
__global__ __launch_bounds__(THREADS) void cn_key_output(const unsigned char* state, const unsigned int* root, const unsigned int* comp_rank,
                                                         const unsigned int* pos, int64_t n, unsigned long long* out, unsigned long long* n_selected) {
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (int64_t)gridDim.x * blockDim.x) {
        if (state[u] == 1) {
            out[u] = ((unsigned long long)comp_rank[root[u]] << 32) | pos[u];
            atomicAdd(n_selected, 1ull);
        } else {
            out[u] = ~0ull;
        }
    }
}

__global__ __launch_bounds__(THREADS) void cn_records(const unsigned int* out_nodes, int64_t n_selected, const unsigned long long* node_key,
                                                      const unsigned int* node_row, const unsigned int* root, const CompStats* st,
                                                      const int64_t* cc, const double* p, const double* q, int64_t res, int64_t r0,
                                                      fhx_cni_record* rec) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_selected; t += (int64_t)gridDim.x * blockDim.x) {
        const unsigned int u = out_nodes[t];
        const unsigned long long key = node_key[u];
        const unsigned int row = node_row[u];
        const CompStats s = st[root[u]];
        fhx_cni_record r{};
        r.chr = (int32_t)(key >> (2 * BIN_BITS));
        r.n_lo = (int64_t)((key >> BIN_BITS) & BIN_MASK) * res + r0;
        r.n_hi = (int64_t)(key & BIN_MASK) * res + r0;
        r.cc = cc[row];
        r.p = p[row];
        r.q = q[row];
        r.box_min_lo = (int64_t)s.min_lo * res + r0;
        r.box_max_lo = (int64_t)s.max_lo * res + r0;
        r.box_min_hi = (int64_t)s.min_hi * res + r0;
        r.box_max_hi = (int64_t)s.max_hi * res + r0;
        r.sum_cc = s.sum_cc;
        r.box_cells = (int64_t)s.have;
        r.component_size = (int64_t)s.size;
        r.first_row = (int64_t)row;
        rec[t] = r;
    }
}

}  // namespace cnd

// ===================================================================================================================
struct fhx_cni {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    fhx_ctx* sorter = nullptr;
    int64_t rows = 0, res = 0, lattice_r = 0;
    int64_t* d_cc = nullptr;
    double* d_q = nullptr;
    double* d_p = nullptr;
    int64_t n_nodes = 0;
    unsigned long long* d_node_key = nullptr;
    unsigned int* d_node_row = nullptr;
    // results of the last run
    std::vector<fhx_cni_record> out;
    fhx_cni_info info{};
};

namespace {

int cfail(fhx_cni* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define CN_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return cfail(cn, FHX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
void cfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

inline int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + cnd::THREADS - 1) / cnd::THREADS, 256 * 16)); }

// RAII for the many device temporaries of one call
struct DevPool {
    std::vector<void*> ptrs;
    ~DevPool() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t get(T** p, size_t count) {
        hipError_t e = hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

}  // namespace

extern "C" {

int fhx_cni_create(int device, fhx_cni** out) {
    if (!out) return FHX_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return FHX_ERR_NO_DEVICE;
    fhx_cni* cn = new fhx_cni();
    cn->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&cn->stream, hipStreamNonBlocking) != hipSuccess ||
        fhx_create(device, &cn->sorter) != FHX_OK) {
        delete cn;
        return FHX_ERR_HIP;
    }
    *out = cn;
    return FHX_OK;
}

void fhx_cni_destroy(fhx_cni* cn) {
    if (!cn) return;
    (void)hipSetDevice(cn->device);
    if (cn->stream) (void)hipStreamSynchronize(cn->stream);
    cfree(cn->d_cc);
    cfree(cn->d_q);
    cfree(cn->d_p);
    cfree(cn->d_node_key);
    cfree(cn->d_node_row);
    if (cn->sorter) fhx_destroy(cn->sorter);
    if (cn->stream) (void)hipStreamDestroy(cn->stream);
    delete cn;
}

const char* fhx_cni_last_error(const fhx_cni* cn) { return cn ? cn->err.c_str() : "null context"; }

int fhx_cni_load(fhx_cni* cn, const int32_t* chr, const int64_t* n1, const int64_t* n2, const int64_t* cc, const double* p,
                 const double* q, int64_t rows, int64_t bin_size, int64_t* n_nodes) {
    if (!cn || rows < 0 || bin_size <= 0 || (rows > 0 && (!chr || !n1 || !n2 || !cc || !p || !q))) return FHX_ERR_ARG;
    if (rows >= (1ll << 32)) return cfail(cn, FHX_ERR_UNSUPPORTED, "more than 2^32 rows");
    CN_HIP(hipSetDevice(cn->device));
    cfree(cn->d_cc);
    cfree(cn->d_q);
    cfree(cn->d_p);
    cfree(cn->d_node_key);
    cfree(cn->d_node_row);
    cn->out.clear();
    cn->rows = rows;
    cn->res = bin_size;
    cn->n_nodes = 0;
    if (n_nodes) *n_nodes = 0;
    if (rows == 0) return FHX_OK;
    // the lattice: every numerator must sit at the same offset r modulo the bin size
    int64_t r = n1[0] % bin_size;
    if (r < 0) r += bin_size;
    cn->lattice_r = r;
    DevPool pool;
    int32_t* d_chr = nullptr;
    int64_t *d_n1 = nullptr, *d_n2 = nullptr;
    unsigned long long *keys = nullptr, *skeys = nullptr, *tile_off = nullptr;
    unsigned int *perm = nullptr, *tile_cnt = nullptr, *d_bad = nullptr;
    CN_HIP(pool.get(&d_chr, rows));
    CN_HIP(pool.get(&d_n1, rows));
    CN_HIP(pool.get(&d_n2, rows));
    CN_HIP(pool.get(&keys, rows));
    CN_HIP(pool.get(&skeys, rows));
    CN_HIP(pool.get(&perm, rows));
    CN_HIP(pool.get(&d_bad, 4));
    CN_HIP(hipMalloc(&cn->d_cc, (size_t)rows * 8));
    CN_HIP(hipMalloc(&cn->d_q, (size_t)rows * 8));
    CN_HIP(hipMalloc(&cn->d_p, (size_t)rows * 8));
    CN_HIP(hipMemcpyAsync(d_chr, chr, (size_t)rows * 4, hipMemcpyHostToDevice, cn->stream));
    CN_HIP(hipMemcpyAsync(d_n1, n1, (size_t)rows * 8, hipMemcpyHostToDevice, cn->stream));
    CN_HIP(hipMemcpyAsync(d_n2, n2, (size_t)rows * 8, hipMemcpyHostToDevice, cn->stream));
    CN_HIP(hipMemcpyAsync(cn->d_cc, cc, (size_t)rows * 8, hipMemcpyHostToDevice, cn->stream));
    CN_HIP(hipMemcpyAsync(cn->d_q, q, (size_t)rows * 8, hipMemcpyHostToDevice, cn->stream));
    CN_HIP(hipMemcpyAsync(cn->d_p, p, (size_t)rows * 8, hipMemcpyHostToDevice, cn->stream));
    CN_HIP(hipMemsetAsync(d_bad, 0, 16, cn->stream));
    hipLaunchKernelGGL(cnd::cn_keys, dim3(grid_of(rows)), dim3(cnd::THREADS), 0, cn->stream, rows, d_chr, d_n1, d_n2, bin_size, r, keys, d_bad);
    CN_HIP(hipGetLastError());
    unsigned int bad = 0;
    CN_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, cn->stream));
    CN_HIP(hipStreamSynchronize(cn->stream));
    if (bad)
        return cfail(cn, FHX_ERR_UNSUPPORTED, std::to_string(bad) + " rows are off the bin lattice (int(mid + res/2) must be congruent to " +
                     std::to_string(r) + " modulo the resolution, bins < 2^24, chromosome ids < 65536)");
    int rc = fhx_sort_u64(cn->sorter, keys, rows, skeys, perm);
    if (rc != FHX_OK) return cfail(cn, rc, std::string("sort: ") + fhx_last_error(cn->sorter));
    const int64_t tiles = (rows + fhxscan::TILE - 1) / fhxscan::TILE;
    CN_HIP(pool.get(&tile_cnt, tiles));
    CN_HIP(pool.get(&tile_off, tiles + 1));
    hipLaunchKernelGGL(fhxscan::count_heads, dim3((unsigned)tiles), dim3(fhxscan::THREADS), 0, cn->stream, skeys, rows, tile_cnt);
    hipLaunchKernelGGL(fhxscan::scan_tiles, dim3(1), dim3(fhxscan::THREADS), 0, cn->stream, tile_cnt, tiles, tile_off, tile_off + tiles);
    CN_HIP(hipGetLastError());
    unsigned long long n = 0;
    CN_HIP(hipMemcpyAsync(&n, tile_off + tiles, 8, hipMemcpyDeviceToHost, cn->stream));
    CN_HIP(hipStreamSynchronize(cn->stream));
    CN_HIP(hipMalloc(&cn->d_node_key, (size_t)n * 8));
    CN_HIP(hipMalloc(&cn->d_node_row, (size_t)n * 4));
    hipLaunchKernelGGL(cnd::cn_emit_nodes, dim3((unsigned)tiles), dim3(fhxscan::THREADS), 0, cn->stream, skeys, perm, rows, tile_off,
                       cn->d_node_key, cn->d_node_row);
    CN_HIP(hipGetLastError());
    CN_HIP(hipStreamSynchronize(cn->stream));
    cn->n_nodes = (int64_t)n;
    if (n_nodes) *n_nodes = cn->n_nodes;
    return FHX_OK;
}

int fhx_cni_run(fhx_cni* cn, int32_t connectivity, int32_t top_percent, int32_t neighborhood, int32_t sort_order, fhx_cni_info* info_out) {
    if (!cn || (connectivity != 8 && connectivity != 4) || neighborhood < 0 || (sort_order != 0 && sort_order != 1)) return FHX_ERR_ARG;
    if (top_percent <= 0)
        return cfail(cn, FHX_ERR_UNSUPPORTED, "-p 0 depends on CPython's set iteration order (CombineNearbyInteraction.py:417-437); "
                                              "-p must be in 1..100 (values > 100 select nothing in the reference)");
    CN_HIP(hipSetDevice(cn->device));
    cn->out.clear();
    fhx_cni_info info{};
    info.rows = cn->rows;
    info.nodes = cn->n_nodes;
    const int64_t n = cn->n_nodes;
    if (n == 0 || top_percent > 100) {                    // none of the reference's three branches runs for -p > 100
        cn->info = info;
        if (info_out) *info_out = info;
        return FHX_OK;
    }
    DevPool pool;
    unsigned int *parent = nullptr, *uf_parent = nullptr, *order_a = nullptr, *order_b = nullptr, *perm = nullptr, *pos = nullptr, *d_seg = nullptr, *d_limit = nullptr;
    unsigned long long *k_in = nullptr, *k_out = nullptr, *d_counter = nullptr;
    cnd::CompStats* st = nullptr;
    unsigned char* state = nullptr;
    CN_HIP(pool.get(&parent, n));                         // root of every node, after cn_flatten
    CN_HIP(pool.get(&uf_parent, n));
    CN_HIP(pool.get(&st, n));
    CN_HIP(pool.get(&order_a, n));
    CN_HIP(pool.get(&order_b, n));
    CN_HIP(pool.get(&perm, n));
    CN_HIP(pool.get(&pos, n));
    CN_HIP(pool.get(&d_seg, n));
    CN_HIP(pool.get(&d_limit, n));
    CN_HIP(pool.get(&k_in, n));
    CN_HIP(pool.get(&k_out, n));
    CN_HIP(pool.get(&d_counter, 4));
    CN_HIP(pool.get(&state, n));
    const dim3 g(grid_of(n)), b(cnd::THREADS);
    // components
    hipLaunchKernelGGL(cnd::cn_init_stats, g, b, 0, cn->stream, st, uf_parent, n);
    hipLaunchKernelGGL(cnd::cn_union, g, b, 0, cn->stream, cn->d_node_key, n, (int)connectivity, uf_parent);
    hipLaunchKernelGGL(cnd::cn_flatten, g, b, 0, cn->stream, uf_parent, n, parent);
    hipLaunchKernelGGL(cnd::cn_stats, g, b, 0, cn->stream, cn->d_node_key, cn->d_node_row, parent, cn->d_cc, n, st);
    if (cn->lattice_r == 0)                               // integer cells only equal the float keys when bins are whole numbers
        hipLaunchKernelGGL(cnd::cn_box_count, dim3(grid_of(n * 64)), b, 0, cn->stream, cn->d_node_key, parent, n, st);
    CN_HIP(hipGetLastError());
    // ranking: stable sorts by -CC, then q (or -q), then root, on top of the (chr, lo, hi) key order
    hipLaunchKernelGGL(cnd::cn_key_cc, g, b, 0, cn->stream, cn->d_node_row, cn->d_cc, n, k_in);
    CN_HIP(hipStreamSynchronize(cn->stream));
    int rc = fhx_sort_u64(cn->sorter, k_in, n, k_out, perm);
    if (rc != FHX_OK) return cfail(cn, rc, std::string("sort: ") + fhx_last_error(cn->sorter));
    hipLaunchKernelGGL(cnd::cn_compose, g, b, 0, cn->stream, (const unsigned int*)nullptr, perm, n, order_a);
    hipLaunchKernelGGL(cnd::cn_key_q, g, b, 0, cn->stream, order_a, cn->d_node_row, cn->d_q, (int)sort_order, n, k_in);
    CN_HIP(hipStreamSynchronize(cn->stream));
    rc = fhx_sort_u64(cn->sorter, k_in, n, k_out, perm);
    if (rc != FHX_OK) return cfail(cn, rc, std::string("sort: ") + fhx_last_error(cn->sorter));
    hipLaunchKernelGGL(cnd::cn_compose, g, b, 0, cn->stream, order_a, perm, n, order_b);
    hipLaunchKernelGGL(cnd::cn_key_root, g, b, 0, cn->stream, order_b, parent, n, k_in);
    CN_HIP(hipStreamSynchronize(cn->stream));
    rc = fhx_sort_u64(cn->sorter, k_in, n, k_out, perm);
    if (rc != FHX_OK) return cfail(cn, rc, std::string("sort: ") + fhx_last_error(cn->sorter));
    hipLaunchKernelGGL(cnd::cn_compose, g, b, 0, cn->stream, order_b, perm, n, order_a);       // final order
    hipLaunchKernelGGL(cnd::cn_positions, g, b, 0, cn->stream, order_a, n, pos);
    CN_HIP(hipGetLastError());
    // segments, candidate limits
    CN_HIP(hipMemsetAsync(d_counter, 0, 4 * 8, cn->stream));
    hipLaunchKernelGGL(cnd::cn_seg_starts, g, b, 0, cn->stream, order_a, parent, n, d_seg, d_counter + 0);
    hipLaunchKernelGGL(cnd::cn_limits, g, b, 0, cn->stream, parent, order_a, cn->d_node_row, cn->d_q, st, d_seg, n, (int)top_percent,
                       (int)sort_order, d_limit, d_counter + 1);
    hipLaunchKernelGGL(cnd::cn_pick_init, g, b, 0, cn->stream, parent, pos, d_seg, d_limit, n, state);
    CN_HIP(hipGetLastError());
    int rounds = 0;
    while (true) {
        CN_HIP(hipMemsetAsync(d_counter + 2, 0, 8, cn->stream));
        hipLaunchKernelGGL(cnd::cn_pick_round, g, b, 0, cn->stream, cn->d_node_key, parent, pos, n, (int)neighborhood, state, d_counter + 2);
        unsigned long long left = 0;
        CN_HIP(hipMemcpyAsync(&left, d_counter + 2, 8, hipMemcpyDeviceToHost, cn->stream));
        CN_HIP(hipStreamSynchronize(cn->stream));
        ++rounds;
        if (left == 0) break;
        if (rounds > 1000000) return cfail(cn, FHX_ERR_HIP, "pick rounds do not converge");
    }
    // output order of the components: first row, then (chromosome, size descending) - two stable sorts over the roots
    hipLaunchKernelGGL(cnd::cn_key_first_row, g, b, 0, cn->stream, parent, st, n, k_in);
    CN_HIP(hipStreamSynchronize(cn->stream));
    rc = fhx_sort_u64(cn->sorter, k_in, n, k_out, perm);
    if (rc != FHX_OK) return cfail(cn, rc, std::string("sort: ") + fhx_last_error(cn->sorter));
    hipLaunchKernelGGL(cnd::cn_compose, g, b, 0, cn->stream, (const unsigned int*)nullptr, perm, n, order_b);
    hipLaunchKernelGGL(cnd::cn_key_chr_size, g, b, 0, cn->stream, order_b, cn->d_node_key, parent, st, n, k_in);
    CN_HIP(hipStreamSynchronize(cn->stream));
    rc = fhx_sort_u64(cn->sorter, k_in, n, k_out, perm);
    if (rc != FHX_OK) return cfail(cn, rc, std::string("sort: ") + fhx_last_error(cn->sorter));
    unsigned long long h_counts[3] = {0, 0, 0};
    CN_HIP(hipMemcpy(h_counts, d_counter, 16, hipMemcpyDeviceToHost));
    const int64_t n_comp = (int64_t)h_counts[0];
    unsigned int* comps_sorted = d_limit;                 // limits are no longer needed
    hipLaunchKernelGGL(cnd::cn_compose, g, b, 0, cn->stream, order_b, perm, n, comps_sorted);
    unsigned int* comp_rank = d_seg;                      // nor are the segment starts
    hipLaunchKernelGGL(cnd::cn_comp_ranks, dim3(grid_of(n_comp)), b, 0, cn->stream, comps_sorted, n_comp, comp_rank);
    // picked nodes in output order
    CN_HIP(hipMemsetAsync(d_counter + 3, 0, 8, cn->stream));
    hipLaunchKernelGGL(cnd::cn_key_output, g, b, 0, cn->stream, state, parent, comp_rank, pos, n, k_in, d_counter + 3);
    CN_HIP(hipGetLastError());
    unsigned long long n_sel = 0;
    CN_HIP(hipMemcpyAsync(&n_sel, d_counter + 3, 8, hipMemcpyDeviceToHost, cn->stream));
    CN_HIP(hipStreamSynchronize(cn->stream));
    rc = fhx_sort_u64(cn->sorter, k_in, n, k_out, perm);
    if (rc != FHX_OK) return cfail(cn, rc, std::string("sort: ") + fhx_last_error(cn->sorter));
    cn->out.resize((size_t)n_sel);
    if (n_sel) {
        fhx_cni_record* d_rec = nullptr;
        CN_HIP(pool.get(&d_rec, (size_t)n_sel));
        hipLaunchKernelGGL(cnd::cn_records, dim3(grid_of((int64_t)n_sel)), b, 0, cn->stream, perm, (int64_t)n_sel, cn->d_node_key,
                           cn->d_node_row, parent, st, cn->d_cc, cn->d_p, cn->d_q, cn->res, cn->lattice_r, d_rec);
        CN_HIP(hipGetLastError());
        CN_HIP(hipMemcpyAsync(cn->out.data(), d_rec, (size_t)n_sel * sizeof(fhx_cni_record), hipMemcpyDeviceToHost, cn->stream));
        CN_HIP(hipStreamSynchronize(cn->stream));
    }
    info.components = n_comp;
    info.selected = (int64_t)n_sel;
    info.pick_rounds = rounds;
    info.largest_component = (int64_t)h_counts[1];
    cn->info = info;
    if (info_out) *info_out = info;
    return FHX_OK;
}

int fhx_cni_get_records(const fhx_cni* cn, fhx_cni_record* out, int64_t capacity, int64_t* n_out) {
    if (!cn) return FHX_ERR_ARG;
    if (n_out) *n_out = (int64_t)cn->out.size();
    if (out) {
        if (capacity < (int64_t)cn->out.size()) return FHX_ERR_ARG;
        std::copy(cn->out.begin(), cn->out.end(), out);
    }
    return FHX_OK;
}

}  // extern "C"
