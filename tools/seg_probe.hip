#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <cstdint>
#include <cstdio>
#include <vector>
int main() {
    const int T = 48; const size_t per = 163840, n = per * T;
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % 10000000); }
    uint32_t *ka, *kb, *va, *vb;
    hipMalloc(&ka, n * 4); hipMalloc(&kb, n * 4); hipMalloc(&va, n * 4); hipMalloc(&vb, n * 4);
    hipMemcpy(ka, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(va, h.data(), n * 4, hipMemcpyHostToDevice);
    size_t tb = 0;
    rocprim::radix_sort_pairs(nullptr, tb, ka, kb, va, vb, per, 0u, 24u, 0);
    const int NS = 8;
    hipStream_t st[NS]; void* temp[NS];
    for (int i = 0; i < NS; ++i) { hipStreamCreate(&st[i]); hipMalloc(&temp[i], tb); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ns : {1, 8}) {
        for (int rep = 0; rep < 3; ++rep) {
            if (rep == 1) { hipDeviceSynchronize(); hipEventRecord(e0, 0); }
            for (int t = 0; t < T; ++t) {
                size_t b = tb;
                rocprim::radix_sort_pairs(temp[t % ns], b, ka + t * per, kb + t * per, va + t * per, vb + t * per, per, 0u, 24u, ns == 1 ? 0 : st[t % ns]);
            }
        }
        hipDeviceSynchronize(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("48 per-table sorts (24 bits) on %d stream(s): %.1f us per round\n", ns, ms * 1000 / 2);
    }
    // segmented sort: 48 segments
    std::vector<uint32_t> offs(T + 1); for (int t = 0; t <= T; ++t) offs[t] = (uint32_t)(t * per);
    uint32_t* d_off; hipMalloc(&d_off, (T + 1) * 4); hipMemcpy(d_off, offs.data(), (T + 1) * 4, hipMemcpyHostToDevice);
    size_t sb = 0;
    rocprim::segmented_radix_sort_pairs(nullptr, sb, ka, kb, va, vb, n, T, d_off, d_off + 1, 0u, 24u, 0);
    void* stemp; hipMalloc(&stemp, sb);
    for (int rep = 0; rep < 3; ++rep) {
        if (rep == 1) { hipDeviceSynchronize(); hipEventRecord(e0, 0); }
        rocprim::segmented_radix_sort_pairs(stemp, sb, ka, kb, va, vb, n, T, d_off, d_off + 1, 0u, 24u, 0);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("segmented_radix_sort_pairs, 48 segments x 163840, 24 bits: %.1f us\n", ms * 1000 / 2);
    return 0;
}
