// one-off: rocPRIM onesweep config sweep for 7.86 M (uint32 key, uint32 value) pairs, 30 key bits
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdint>
#include <cstdio>
#include <vector>
template <int BS, int IPT, int BITS>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, BITS>>;
template <class C>
float run(const char* name, uint32_t* ka, uint32_t* kb, uint32_t* va, uint32_t* vb, size_t n, int bits) {
    size_t tb = 0;
    rocprim::radix_sort_pairs<C>(nullptr, tb, ka, kb, va, vb, n, 0u, (unsigned)bits, 0);
    void* temp; hipMalloc(&temp, tb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs<C>(temp, tb, ka, kb, va, vb, n, 0u, (unsigned)bits, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs<C>(temp, tb, ka, kb, va, vb, n, 0u, (unsigned)bits, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s %.1f us per sort (temp %zu B)\n", name, ms * 1000 / 20, tb);
    hipFree(temp);
    return ms;
}
int main() {
    const size_t n = 7864320; const int bits = 30;
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)((i / 163840) << 24) | (uint32_t)(s % 10000000); }
    uint32_t *ka, *kb, *va, *vb;
    hipMalloc(&ka, n * 4); hipMalloc(&kb, n * 4); hipMalloc(&va, n * 4); hipMalloc(&vb, n * 4);
    hipMemcpy(ka, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(va, h.data(), n * 4, hipMemcpyHostToDevice);
    run<rocprim::default_config>("default", ka, kb, va, vb, n, bits);
    run<Cfg<256, 8, 8>>("256x8 r8", ka, kb, va, vb, n, bits);
    run<Cfg<256, 12, 8>>("256x12 r8", ka, kb, va, vb, n, bits);
    run<Cfg<256, 16, 8>>("256x16 r8", ka, kb, va, vb, n, bits);
    run<Cfg<256, 24, 8>>("256x24 r8", ka, kb, va, vb, n, bits);
    run<Cfg<128, 16, 7>>("128x16 r7", ka, kb, va, vb, n, bits);
    run<Cfg<256, 16, 6>>("256x16 r6", ka, kb, va, vb, n, bits);
    return 0;
}
