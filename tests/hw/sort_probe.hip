// Timing probe: hipcub radix sort of (u64 key with 40 significant bits, u32 value) pairs, as a planning step
// candidate (sort-based de-duplication of alignments).  Build: hipcc --offload-arch=gfx950 -O3 sort_probe.hip -o sort_probe
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 33542107ull;
    const int bits = argc > 2 ? atoi(argv[2]) : 40;
    std::vector<unsigned long long> h(n); std::vector<unsigned> v(n);
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x & ((1ull << bits) - 1); v[i] = (unsigned)i; }
    unsigned long long *k0, *k1; unsigned *v0, *v1;
    hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    hipMemcpy(k0, h.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(v0, v.data(), n * 4, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0; void* tmp = nullptr;
    hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits);
    hipMalloc(&tmp, tmp_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("n=%zu bits=%d sort pairs: %.3f ms (temp %.1f MB)\n", n, bits, ms, tmp_bytes / 1e6);
    }
    std::vector<unsigned long long> o(n); hipMemcpy(o.data(), k1, n * 8, hipMemcpyDeviceToHost);
    bool ok = true; for (size_t i = 1; i < n; ++i) if (o[i - 1] > o[i]) { ok = false; break; }
    printf("sorted: %d\n", (int)ok);
    return ok ? 0 : 1;
}
