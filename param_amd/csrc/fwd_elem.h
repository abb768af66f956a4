// param_amd/csrc/fwd_elem.h -- element types and row-load helpers of the forward kernels (embbag_fwd.hip).  gfx950 only.
#pragma once

#include "common.h"

namespace pm {
namespace fwd {

struct bf16_t { uint16_t v; };
struct f16_t { uint16_t v; };

template <typename WT> struct Elem;
template <> struct Elem<float> {
    static constexpr int kVec = 4;
    __device__ static __forceinline__ void widen(const u32x4& raw, float (&f)[4]) {
        f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y);
        f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
    }
};
template <> struct Elem<bf16_t> {
    static constexpr int kVec = 8;
    __device__ static __forceinline__ void widen(const u32x4& raw, float (&f)[8]) {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
};
template <> struct Elem<f16_t> {
    static constexpr int kVec = 8;
    __device__ static __forceinline__ void widen(const u32x4& raw, float (&f)[8]) {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(w[i] & 0xffffu)));
            f[2 * i + 1] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(w[i] >> 16)));
        }
    }
};

__device__ __forceinline__ u32x4 load16(const char* p, bool nt) {
    const PM_GLOBAL u32x4* q = as_global<u32x4>(p);   // global_load_dwordx4, not flat_load (common.h)
    return nt ? __builtin_nontemporal_load(q) : *q;
}

// byte offset of row r: staged indices are int32 (rows[t] < 2^31, the caller's contract) and a row is < 2^31 bytes, so the
// product is ONE 32 x 32 -> 64-bit multiply (v_mad_u64_u32); as int64 x int64 it is a multiply-add, two multiplies and an add
template <bool ST>
__device__ __forceinline__ int64_t row_offset(int64_t r, int64_t row_bytes) {
    if (ST) return static_cast<int64_t>(static_cast<uint64_t>(static_cast<uint32_t>(r)) * static_cast<uint32_t>(row_bytes));
    return r * row_bytes;
}

}  // namespace fwd
}  // namespace pm
