// fhx_cpus.hpp - how many host threads are worth starting.
//
// std::thread::hardware_concurrency() reports the machine (256 on the GPU box) while a container may be allowed a fraction
// of it (cgroup cpu.max: 16 CPUs there): 256 workers on a 16-CPU quota are throttled in bursts and trash each other's
// caches.  The count used by the reader, the writers and the Huffman-code builder is the smallest of: hardware threads,
// the affinity mask, the cgroup quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us); FHX_THREADS overrides.
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace fhx {

inline int usable_cpus() {
    if (const char* e = std::getenv("FHX_THREADS")) {
        const int v = std::atoi(e);
        if (v > 0) return v;
    }
    long n = (long)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const long a = CPU_COUNT(&set);
        if (a > 0) n = std::min(n, a);
    }
    auto quota = [](const char* path_quota, const char* path_period) -> long {
        long q = -1, p = 100000;
        if (std::FILE* f = std::fopen(path_quota, "r")) {
            char word[32] = {0};
            if (path_period == nullptr) {                       // v2: "max 100000" or "<quota> <period>"
                if (std::fscanf(f, "%31s %ld", word, &p) >= 1 && word[0] != 'm') q = std::atol(word);
            } else if (std::fscanf(f, "%ld", &q) != 1) {
                q = -1;
            }
            std::fclose(f);
        }
        if (path_period)
            if (std::FILE* f = std::fopen(path_period, "r")) {
                if (std::fscanf(f, "%ld", &p) != 1) p = 100000;
                std::fclose(f);
            }
        return (q > 0 && p > 0) ? (q + p - 1) / p : -1;
    };
    long c = quota("/sys/fs/cgroup/cpu.max", nullptr);
    if (c <= 0) c = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (c > 0) n = std::min(n, c);
    return (int)std::max(1l, n);
}

}  // namespace fhx
