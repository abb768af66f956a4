// param_amd/csrc/embbag_fwd_split.hip -- EmbeddingBag(sum) forward for FEW, LONG bags: one workgroup per bag.
//
// The default forward (embbag_fwd.hip) gives a bag to ONE lane group, which walks its lookups sequentially: right
// when there are many bags (the benchmark: 393 216 bags of 20), hopeless when there are few long ones (inference-style
// requests: a handful of bags of thousands of lookups leave the chip idle and each bag is a serial chain).  Here the
// bag is split: the NG = 256 / G lane groups of a workgroup each pool every NG-th lookup of the bag into fp32
// registers (UNROLL row loads in flight per lane, indices staged through LDS in chunks), and the partial sums are
// combined
//   1. across the lane groups of a wave with a butterfly of wavefront shuffles (__shfl_xor over lane offsets
//      G, 2G, ... 32), then
//   2. across the 4 waves through LDS, in wave order, by the first lane group, which writes the pooled row.
// The order of the additions is fixed (deterministic) but it is not the sequential order: results agree with the
// sequential sum to fp32 rounding (tests: 1e-5 relative to sum |row|), not bit for bit -- which is why this is a separate,
// explicitly requested entry point (pm_embbag_fwd_split) and the default stays the bit-exact kernel.
#include "common.h"

namespace pm {
namespace {

constexpr int kIdxChunk = 2048;  // indices of the bag staged in LDS per round

struct sbf16_t { uint16_t v; };
struct sf16_t { uint16_t v; };
template <typename WT> struct SElem;
template <> struct SElem<float> {
    static constexpr int kVec = 4;
    __device__ static __forceinline__ void widen(const u32x4& raw, float (&f)[4]) {
        f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y); f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
    }
};
template <> struct SElem<sbf16_t> {
    static constexpr int kVec = 8;
    __device__ static __forceinline__ void widen(const u32x4& raw, float (&f)[8]) {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
};
template <> struct SElem<sf16_t> {
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

template <typename WT, int G, bool WEIGHTED>
__global__ void __launch_bounds__(kBlock) embbag_fwd_split_kernel(const KParams p) {
    constexpr int VEC = SElem<WT>::kVec;
    constexpr int NG = kBlock / G;
    constexpr int UNROLL = 4;
    constexpr int ES = 16 / VEC;
    __shared__ int32_t s_idx[kIdxChunk];
    __shared__ float s_w[WEIGHTED ? kIdxChunk : 1];
    __shared__ float s_part[(kBlock / kWave) * G * VEC];   // one partial row slice per wave (one column pass at a time)

    const int t = blockIdx.x / static_cast<int>(p.bag_count);
    const int64_t b = p.bag_begin + blockIdx.x % p.bag_count;
    const int64_t gbag = static_cast<int64_t>(t) * p.B + b;
    const int64_t s = bag_start_or_end(p, gbag);
    const int64_t e = bag_start_or_end(p, gbag + 1);
    const int D = p.dims[t];
    const int64_t row_bytes = static_cast<int64_t>(D) * ES;
    const char* W = reinterpret_cast<const char*>(p.tables[t]);
    float* orow = p.io + p.out_offsets[t] + b * p.out_stride;
    const int gid = threadIdx.x / G;
    const int lig = threadIdx.x % G;
    const int wave = threadIdx.x / kWave;

    for (int c = lig * VEC; c - lig * VEC < D; c += G * VEC) {   // column passes: all lanes take every pass (barriers inside)
        const bool col = c < D;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
        for (int64_t base = s; base < e; base += kIdxChunk) {
            const int n = (e - base) < kIdxChunk ? static_cast<int>(e - base) : kIdxChunk;
            __syncthreads();                                  // the previous chunk (or pass) is done with s_idx
            for (int i = threadIdx.x; i < n; i += kBlock) {
                s_idx[i] = static_cast<int32_t>(load_index(p.indices, base + i, p.idx64));
                if (WEIGHTED) s_w[i] = p.psw[base + i];
            }
            __syncthreads();
            if (col) {
                const char* Wc = W + static_cast<int64_t>(c) * ES;
                // this group's lookups of the chunk: gid, gid + NG, ... ; UNROLL of them in flight
                for (int j = gid; j < n; j += NG * UNROLL) {
                    u32x4 raw[UNROLL];
                    float w[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const int jj = j + u * NG;
                        if (jj < n) {
                            raw[u] = *as_global<u32x4>(Wc + static_cast<int64_t>(s_idx[jj]) * row_bytes);
                            if (WEIGHTED) w[u] = s_w[jj];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        if (j + u * NG < n) {
                            float f[VEC];
                            SElem<WT>::widen(raw[u], f);
#pragma unroll
                            for (int k = 0; k < VEC; ++k) acc[k] = WEIGHTED ? fmaf(w[u], f[k], acc[k]) : acc[k] + f[k];
                        }
                    }
                }
            }
        }
        // 1. lane groups of one wave: butterfly of wavefront shuffles over the group index bits
#pragma unroll
        for (int off = G; off < kWave; off <<= 1) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] + __shfl_xor(acc[k], off, kWave);
        }
        // 2. the waves: through LDS, summed in wave order by the first lane group
        if ((threadIdx.x % kWave) < G) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) s_part[(wave * G + lig) * VEC + k] = acc[k];
        }
        __syncthreads();
        if (gid == 0 && col) {
            float tot[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) tot[k] = s_part[lig * VEC + k];
            for (int wv = 1; wv < kBlock / kWave; ++wv) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) tot[k] = tot[k] + s_part[(wv * G + lig) * VEC + k];
            }
            f32x4* o4 = reinterpret_cast<f32x4*>(orow + c);
#pragma unroll
            for (int k = 0; k < VEC; k += 4) {
                f32x4 v = {tot[k], tot[k + 1], tot[k + 2], tot[k + 3]};
                __builtin_nontemporal_store(v, o4 + k / 4);
            }
        }
        __syncthreads();                                      // s_part is rewritten by the next column pass
    }
}

template <typename WT, int G>
hipError_t launch_w(const KParams& p, hipStream_t stream) {
    const int64_t grid = static_cast<int64_t>(p.T) * p.bag_count;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    if (p.psw)
        hipLaunchKernelGGL((embbag_fwd_split_kernel<WT, G, true>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, p);
    else
        hipLaunchKernelGGL((embbag_fwd_split_kernel<WT, G, false>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, p);
    return hipGetLastError();
}

template <typename WT>
hipError_t launch_g(const KParams& p, int max_dim, hipStream_t stream) {
    switch (group_lanes(max_dim, SElem<WT>::kVec)) {
        case 8: return launch_w<WT, 8>(p, stream);
        case 16: return launch_w<WT, 16>(p, stream);
        case 32: return launch_w<WT, 32>(p, stream);
        default: return launch_w<WT, 64>(p, stream);
    }
}

}  // namespace

hipError_t launch_embbag_fwd_split(const KParams& p, int weight_dtype, int max_dim, hipStream_t stream) {
    switch (weight_dtype) {
        case PM_F32: return launch_g<float>(p, max_dim, stream);
        case PM_BF16: return launch_g<sbf16_t>(p, max_dim, stream);
        default: return launch_g<sf16_t>(p, max_dim, stream);
    }
}

}  // namespace pm
