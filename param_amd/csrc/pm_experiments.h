// param_amd/csrc/pm_experiments.h -- the ONE switch for experiment builds of libparam_amd (round 5).
//
// The product Makefile never defines PM_EXPERIMENTS: every macro below then expands to nothing and the product library has no
// trace buffer, no extra symbol and no extra instruction (tests/test_static_names.py checks the sources for stray switches,
// tests/test_capi_symbols.py the exports).  An experiment build
//     make -C param_amd/csrc EXTRA=-DPM_EXPERIMENTS OBJDIR=$PWD/build/csrc_exp OUT=$PWD/build/libparam_amd_exp.so
// adds per-workgroup phase timestamps (the 100 MHz s_memrealtime counter, one clock for the whole chip) to the kernels that
// carry PM_STAMP marks, a device buffer for them and `pm_experiment_trace` to read it back (tools/r5_sort_trace.py).  Results of
// an experiment build are the product's: stamps only observe.
#pragma once

#ifdef PM_EXPERIMENTS
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace pm {
namespace exp {
constexpr int kTraceSlots = 8;
constexpr int kTraceWgs = 1 << 15;
namespace {      // one buffer per translation unit (the library is not built with relocatable device code)
__device__ unsigned long long g_trace[kTraceWgs * kTraceSlots];
}
__device__ __forceinline__ void stamp(unsigned wg, int slot, bool drain) {
    if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && wg < static_cast<unsigned>(kTraceWgs)) g_trace[wg * kTraceSlots + slot] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace exp
}  // namespace pm
#define PM_STAMP(wg, slot) pm::exp::stamp((wg), (slot), false)
#define PM_STAMP_DRAINED(wg, slot) pm::exp::stamp((wg), (slot), true)
// the translation unit's reader: copies `words` 64-bit words of its trace buffer to `out` (device synchronised first), optionally clears it
#define PM_DEFINE_TRACE_READER(fn)                                                                                                      \
    extern "C" int fn(unsigned long long* out, int words, int clear) {                                                                  \
        const size_t all = sizeof(unsigned long long) * pm::exp::kTraceWgs * pm::exp::kTraceSlots;                                      \
        const size_t bytes = sizeof(unsigned long long) * static_cast<size_t>(words);                                                   \
        if (bytes > all) return -1;                                                                                                     \
        if (hipDeviceSynchronize() != hipSuccess) return -2;                                                                            \
        if (out && bytes && hipMemcpyFromSymbol(out, HIP_SYMBOL(pm::exp::g_trace), bytes) != hipSuccess) return -3;                     \
        if (clear) {                                                                                                                    \
            void* p = nullptr;                                                                                                          \
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(pm::exp::g_trace)) != hipSuccess || hipMemset(p, 0, all) != hipSuccess) return -4;   \
        }                                                                                                                               \
        return 0;                                                                                                                       \
    }
#else
#define PM_STAMP(wg, slot) ((void)0)
#define PM_STAMP_DRAINED(wg, slot) ((void)0)
#define PM_DEFINE_TRACE_READER(fn)
#endif
