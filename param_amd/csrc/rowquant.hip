// param_amd/csrc/rowquant.hip -- row-wise quantisation of pooled embeddings for the quantised all-to-all.
//
// The reference's comms drivers take --bitwidth {2,4,8,16,32}, --quant-a2a-embedding-dim {32,64,128,256} and
// --quant-threshold (comms_utils.py:1788-1806): fp32 payloads are downcast before the exchange and restored after it
// (pytorch_dist_backend.py:48-76 for the open part: fp16 for 16 bits; the all-to-all variant lives in a package that is
// not published, all_to_allv_internal at :273).  The row formats here are the published fbgemm "fused row-wise" ones,
// which torch ships as quantized::embedding_bag_{byte,4bit,2bit}_prepack / _unpack (the parity oracle, oracle/rowquant.py):
//
//   16 bits : [dim x fp16]                                   round-to-nearest-even cast
//    8 bits : [dim x u8][fp32 scale][fp32 bias]              bias = min, scale = (max - min) / 255,
//                                                            code = rint((x - min) * (255 / (max - min + 1e-8)))
//  4/2 bits : [dim*bits/8 x u8][fp16 scale][fp16 bias]       bias = fp16(min), scale = fp16((max - bias) / (2^bits - 1))
//                                                            (1 when that is 0 or its inverse overflows),
//                                                            code = clamp(rint((x - bias) / scale)), low bits first
//   restore : x = fma(code, scale, bias)
//
// HBM-bound byte work: G = 4..64 lanes own a row, a lane two 4-column chunks of it (two 16-byte accesses, each contiguous
// across the lane group), the row minimum / maximum are a shuffle reduction inside the lane group.
#include <algorithm>

#include "common.h"
#include "rowquant.inc"

namespace pm {
namespace {

template <int G>
__device__ __forceinline__ float group_min(float v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, G));
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, G));
    return v;
}

using rq::half_bits;
using rq::half_value;
using rq::load_codes;
using rq::row_bytes;

// BITS = 8, 4, 2: fused row-wise formats.  A lane owns two 4-column chunks of a row, G * 4 columns apart, so that each of
// its two 16-byte accesses is contiguous across the lane group; blockDim = kBlock, kBlock / G row slots per block and
// kRowsPerSlot rows per slot (independent loads first, then the arithmetic and the stores).
constexpr int kRowsPerSlot = 2;    // restore (write-bound)
constexpr int kRowsPerSlotQ = 4;   // quantise (read-bound): more loads in flight per lane

template <int G, int BITS>
__global__ void __launch_bounds__(kBlock) rows_quantize_kernel(const float* __restrict__ src, int64_t n_rows, int dim,
                                                               uint8_t* __restrict__ dst) {
    const int lane = threadIdx.x % G;
    const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G) * kRowsPerSlotQ;
    const int c0 = lane * 4, c1 = (G + lane) * 4;
    const bool has0 = c0 < dim, has1 = c1 < dim;
    f32x4 a[kRowsPerSlotQ], b[kRowsPerSlotQ];
#pragma unroll
    for (int r = 0; r < kRowsPerSlotQ; ++r) {               // (whole lane groups share a row: the shuffles stay inside one)
        const bool live = row0 + r < n_rows;
        const float* p = src + (row0 + r) * dim;
        a[r] = (live && has0) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + c0)) : f32x4{0, 0, 0, 0};
        b[r] = (live && has1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + c1)) : f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int r = 0; r < kRowsPerSlotQ; ++r) {
        const bool live = row0 + r < n_rows;
        float mn, mx;
        rq::row_min_max(a[r], b[r], has0, has1, mn, mx);
        mn = group_min<G>(mn);
        mx = group_max<G>(mx);
        if (!live) continue;
        const float x[8] = {a[r].x, a[r].y, a[r].z, a[r].w, b[r].x, b[r].y, b[r].z, b[r].w};
        rq::quantize_row_share<BITS>(x, mn, mx, lane == 0, has0, has1, c0, c1, dim, dst + (row0 + r) * row_bytes(dim, BITS));
    }
}

template <int G, int BITS>
__global__ void __launch_bounds__(kBlock) rows_dequantize_kernel(const uint8_t* __restrict__ src, int64_t n_rows, int dim,
                                                                 float* __restrict__ dst) {
    const int lane = threadIdx.x % G;
    const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G) * kRowsPerSlot;
    const int c0 = lane * 4, c1 = (G + lane) * 4;
    const bool has0 = c0 < dim, has1 = c1 < dim;
    uint32_t w0[kRowsPerSlot], w1[kRowsPerSlot];
    float scale[kRowsPerSlot], bias[kRowsPerSlot];
#pragma unroll
    for (int r = 0; r < kRowsPerSlot; ++r) {
        const bool live = row0 + r < n_rows;
        const uint8_t* in = src + (row0 + r) * row_bytes(dim, BITS);
        w0[r] = (live && has0) ? load_codes<BITS>(in, c0) : 0u;
        w1[r] = (live && has1) ? load_codes<BITS>(in, c1) : 0u;
        scale[r] = bias[r] = 0.0f;
        if (live) {
            if constexpr (BITS == 8) {
                const float2 sb = *reinterpret_cast<const float2*>(in + dim);
                scale[r] = sb.x;
                bias[r] = sb.y;
            } else {
                const uint32_t sb = *reinterpret_cast<const uint16_t*>(in + dim * BITS / 8) |
                                    (static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(in + dim * BITS / 8 + 2)) << 16);
                scale[r] = half_value(static_cast<uint16_t>(sb & 0xffffu));
                bias[r] = half_value(static_cast<uint16_t>(sb >> 16));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kRowsPerSlot; ++r) {
        if (row0 + r >= n_rows) continue;
        constexpr uint32_t kMask = (1u << BITS) - 1u;
        float* o = dst + (row0 + r) * dim;
        f32x4 y0, y1;
        y0.x = __fmaf_rn(static_cast<float>(w0[r] & kMask), scale[r], bias[r]);
        y0.y = __fmaf_rn(static_cast<float>((w0[r] >> BITS) & kMask), scale[r], bias[r]);
        y0.z = __fmaf_rn(static_cast<float>((w0[r] >> (2 * BITS)) & kMask), scale[r], bias[r]);
        y0.w = __fmaf_rn(static_cast<float>((w0[r] >> (3 * BITS)) & kMask), scale[r], bias[r]);
        y1.x = __fmaf_rn(static_cast<float>(w1[r] & kMask), scale[r], bias[r]);
        y1.y = __fmaf_rn(static_cast<float>((w1[r] >> BITS) & kMask), scale[r], bias[r]);
        y1.z = __fmaf_rn(static_cast<float>((w1[r] >> (2 * BITS)) & kMask), scale[r], bias[r]);
        y1.w = __fmaf_rn(static_cast<float>((w1[r] >> (3 * BITS)) & kMask), scale[r], bias[r]);
        if (has0) __builtin_nontemporal_store(y0, reinterpret_cast<f32x4*>(o + c0));
        if (has1) __builtin_nontemporal_store(y1, reinterpret_cast<f32x4*>(o + c1));
    }
}

// 16 bits: an element-wise cast.  A wave covers kCastStep * 512 consecutive elements per step; a lane's k-th access is 4
// elements at (k * 64 + lane) * 4, so every load and store instruction is contiguous across the wave.
constexpr int kCastStep = 4;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(kBlock) cast_f32_f16_kernel(const float* __restrict__ src, int64_t n, uint16_t* __restrict__ dst) {
    const int lane = threadIdx.x % 64;
    const int64_t waves = static_cast<int64_t>(gridDim.x) * (kBlock / 64);
    constexpr int64_t kSpan = 64 * 4 * kCastStep;
    for (int64_t base = (static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + threadIdx.x / 64) * kSpan; base < n; base += waves * kSpan) {
        f32x4 v[kCastStep];
#pragma unroll
        for (int k = 0; k < kCastStep; ++k) {
            const int64_t e = base + (k * 64 + lane) * 4;
            v[k] = e < n ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + e)) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < kCastStep; ++k) {
            const int64_t e = base + (k * 64 + lane) * 4;
            if (e >= n) continue;
            u32x2 o;
            o.x = half_bits(v[k].x) | (static_cast<uint32_t>(half_bits(v[k].y)) << 16);
            o.y = half_bits(v[k].z) | (static_cast<uint32_t>(half_bits(v[k].w)) << 16);
            __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(dst + e));
        }
    }
}
__global__ void __launch_bounds__(kBlock) cast_f16_f32_kernel(const uint16_t* __restrict__ src, int64_t n, float* __restrict__ dst) {
    const int lane = threadIdx.x % 64;
    const int64_t waves = static_cast<int64_t>(gridDim.x) * (kBlock / 64);
    constexpr int64_t kSpan = 64 * 4 * kCastStep;
    for (int64_t base = (static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + threadIdx.x / 64) * kSpan; base < n; base += waves * kSpan) {
        u32x2 v[kCastStep];
#pragma unroll
        for (int k = 0; k < kCastStep; ++k) {
            const int64_t e = base + (k * 64 + lane) * 4;
            v[k] = e < n ? __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src + e)) : u32x2{0, 0};
        }
#pragma unroll
        for (int k = 0; k < kCastStep; ++k) {
            const int64_t e = base + (k * 64 + lane) * 4;
            if (e >= n) continue;
            __builtin_nontemporal_store(f32x4{half_value(v[k].x & 0xffffu), half_value(v[k].x >> 16), half_value(v[k].y & 0xffffu),
                                              half_value(v[k].y >> 16)}, reinterpret_cast<f32x4*>(dst + e));
        }
    }
}

inline unsigned cast_grid(int64_t n) {
    const int64_t per_block = static_cast<int64_t>(kBlock / 64) * 64 * 4 * kCastStep;
    return static_cast<unsigned>(std::min<int64_t>((n + per_block - 1) / per_block, 256 * 32));
}

int lanes_for(int dim) {
    int g = 4;
    while (g * 8 < dim) g *= 2;
    return g;
}

template <int BITS, bool QUANT>
hipError_t launch_fused(const void* src, int64_t n_rows, int dim, void* dst, hipStream_t stream) {
    const int g = lanes_for(dim);
    const int64_t per_block = static_cast<int64_t>(kBlock / g) * (QUANT ? kRowsPerSlotQ : kRowsPerSlot);
    const int64_t grid = (n_rows + per_block - 1) / per_block;
#define PM_ROWQ(G_)                                                                                                       \
    do {                                                                                                                  \
        if constexpr (QUANT)                                                                                              \
            hipLaunchKernelGGL((rows_quantize_kernel<G_, BITS>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, \
                               static_cast<const float*>(src), n_rows, dim, static_cast<uint8_t*>(dst));                 \
        else                                                                                                              \
            hipLaunchKernelGGL((rows_dequantize_kernel<G_, BITS>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, \
                               static_cast<const uint8_t*>(src), n_rows, dim, static_cast<float*>(dst));                 \
    } while (0)
    switch (g) {
        case 4: PM_ROWQ(4); break;
        case 8: PM_ROWQ(8); break;
        case 16: PM_ROWQ(16); break;
        case 32: PM_ROWQ(32); break;
        default: PM_ROWQ(64); break;
    }
#undef PM_ROWQ
    return hipGetLastError();
}

}  // namespace

int64_t rows_quantized_row_bytes(int dim, int bits) { return row_bytes(dim, bits); }

// dim: multiple of 8, <= 512; bits in {16, 8, 4, 2}
hipError_t launch_rows_quantize(const float* src, int64_t n_rows, int dim, int bits, void* dst, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    switch (bits) {
        case 16: {
            const int64_t n = n_rows * dim;      // (a multiple of 8: every access is a whole 4-element word)
            hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(cast_grid(n)), dim3(kBlock), 0, stream, src, n, static_cast<uint16_t*>(dst));
            return hipGetLastError();
        }
        case 8: return launch_fused<8, true>(src, n_rows, dim, dst, stream);
        case 4: return launch_fused<4, true>(src, n_rows, dim, dst, stream);
        default: return launch_fused<2, true>(src, n_rows, dim, dst, stream);
    }
}

hipError_t launch_rows_dequantize(const void* src, int64_t n_rows, int dim, int bits, float* dst, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    switch (bits) {
        case 16: {
            const int64_t n = n_rows * dim;
            hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(cast_grid(n)), dim3(kBlock), 0, stream, static_cast<const uint16_t*>(src), n, dst);
            return hipGetLastError();
        }
        case 8: return launch_fused<8, false>(src, n_rows, dim, dst, stream);
        case 4: return launch_fused<4, false>(src, n_rows, dim, dst, stream);
        default: return launch_fused<2, false>(src, n_rows, dim, dst, stream);
    }
}

}  // namespace pm
