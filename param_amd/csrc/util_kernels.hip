// param_amd/csrc/util_kernels.hip -- request validation and HBM-speed table initialisation.
#include "common.h"

namespace pm {
namespace {

// One thread per bag: offsets monotone within [0, N]; every index within [0, rows[t]).
// torch raises on such inputs on the CPU path the reference runs
// (train/compute/pt/pytorch_emb.py:40); the hot kernels do not check.
// Block 0 also validates what the kernels assume about the per-table DEVICE arrays, which the host side of the C ABI
// cannot see: rows[t] in [1, 2^31) (staged indices are narrowed to int32), dims[t] a multiple of the 16-byte vector and
// <= max_dim, out_offsets[t] 16-byte aligned, table base pointers 16-byte aligned.
__global__ void __launch_bounds__(kBlock) embbag_check_kernel(const KParams p, int32_t* err, int vec, int max_dim,
                                                              int64_t fixed_pooling, int uniform_dims) {
    const int64_t per_table = p.bag_count;
    const int64_t total = per_table * p.T;
    int bad = 0;
    if (blockIdx.x == 0) {
        for (int t = threadIdx.x; t < p.T; t += kBlock) {
            const int64_t rows = p.rows[t];
            const int d = p.dims[t];
            bad += (rows < 1 || rows >= (1LL << 31)) ? 1 : 0;
            bad += (d < 1 || d % vec != 0 || d > max_dim) ? 1 : 0;
            bad += (p.out_offsets[t] % 4 != 0) ? 1 : 0;
            bad += (reinterpret_cast<uintptr_t>(p.tables[t]) % 16 != 0) ? 1 : 0;
            // quantised output (pm_embbag_fwd_quantized): every pooled vector is one max_dim-element row of the output
            if (uniform_dims) bad += (d != max_dim || p.out_offsets[t] % max_dim != 0) ? 1 : 0;
        }
    }
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int t = static_cast<int>(i / per_table);
        const int64_t b = p.bag_begin + i % per_table;
        const int64_t g = static_cast<int64_t>(t) * p.B + b;
        const int64_t s = bag_start_or_end(p, g);
        const int64_t e = bag_start_or_end(p, g + 1);
        if (s < 0 || e < s || e > p.N) {
            ++bad;
            continue;
        }
        if (fixed_pooling > 0 && (s != g * fixed_pooling || e - s != fixed_pooling)) ++bad;   // the caller's fixed-pooling claim
        const int64_t rows = p.rows[t];
        for (int64_t j = s; j < e; ++j) {
            const int64_t r = load_index(p.indices, j, p.idx64);
            bad += (r < 0 || r >= rows) ? 1 : 0;
        }
    }
    if (bad) atomicAdd(err, bad);
}

// Counter-based generator: element i depends only on (seed, i).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float draw(uint64_t seed, uint64_t i, int dist, float lo, float hi) {
    const uint64_t h = mix64(seed ^ mix64(i));
    const float u1 = (static_cast<uint32_t>(h >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    if (dist == 0) return lo + (hi - lo) * u1;
    const float u2 = (static_cast<uint32_t>(h & 0xffffffu) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = __fsqrt_rn(-2.0f * __logf(u1));
    return lo + hi * rad * __cosf(6.28318530718f * u2);  // Box-Muller, cosine branch
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}

template <int DT>
__global__ void __launch_bounds__(kBlock) fill_random_kernel(void* dst, int64_t count, int dist, float lo,
                                                             float hi, uint64_t seed) {
    constexpr int PER = (DT == PM_F32) ? 4 : 8;  // 16 bytes per thread per step
    const int64_t nvec = count / PER;
    for (int64_t v = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; v < nvec;
         v += static_cast<int64_t>(gridDim.x) * kBlock) {
        float f[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) f[k] = draw(seed, static_cast<uint64_t>(v) * PER + k, dist, lo, hi);
        u32x4 out;
        if (DT == PM_F32) {
            out = u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                             __float_as_uint(f[3])};
        } else {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t a, b;
                if (DT == PM_BF16) {
                    a = f32_to_bf16_rne(f[2 * k]);
                    b = f32_to_bf16_rne(f[2 * k + 1]);
                } else {
                    a = __builtin_bit_cast(uint16_t, static_cast<_Float16>(f[2 * k]));
                    b = __builtin_bit_cast(uint16_t, static_cast<_Float16>(f[2 * k + 1]));
                }
                w[k] = a | (b << 16);
            }
            out = u32x4{w[0], w[1], w[2], w[3]};
        }
        __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(dst) + v);
    }
    // scalar tail
    if (blockIdx.x == 0 && threadIdx.x < count - nvec * PER) {
        const int64_t i = nvec * PER + threadIdx.x;
        const float f = draw(seed, static_cast<uint64_t>(i), dist, lo, hi);
        if (DT == PM_F32) reinterpret_cast<float*>(dst)[i] = f;
        else if (DT == PM_BF16) reinterpret_cast<uint16_t*>(dst)[i] = f32_to_bf16_rne(f);
        else reinterpret_cast<uint16_t*>(dst)[i] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(f));
    }
}

}  // namespace

hipError_t launch_embbag_check(const KParams& p, int32_t* d_err, int vec, int max_dim, int64_t fixed_pooling,
                               int uniform_dims, hipStream_t stream) {
    hipError_t rc = hipMemsetAsync(d_err, 0, sizeof(int32_t), stream);
    if (rc != hipSuccess) return rc;
    const int64_t total = p.bag_count * p.T;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks < 1) blocks = 1;   // the per-table checks run even for an empty batch slice
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(embbag_check_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, p, d_err, vec,
                       max_dim, fixed_pooling, uniform_dims);
    return hipGetLastError();
}

hipError_t launch_fill_random(void* dst, int64_t count, int dtype, int dist, float lo, float hi,
                              uint64_t seed, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int per = (dtype == PM_F32) ? 4 : 8;
    int64_t blocks = (count / per + kBlock - 1) / kBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const dim3 grid(static_cast<unsigned>(blocks));
    switch (dtype) {
        case PM_F32:
            hipLaunchKernelGGL(fill_random_kernel<PM_F32>, grid, dim3(kBlock), 0, stream, dst, count, dist, lo, hi, seed);
            break;
        case PM_BF16:
            hipLaunchKernelGGL(fill_random_kernel<PM_BF16>, grid, dim3(kBlock), 0, stream, dst, count, dist, lo, hi, seed);
            break;
        default:
            hipLaunchKernelGGL(fill_random_kernel<PM_F16>, grid, dim3(kBlock), 0, stream, dst, count, dist, lo, hi, seed);
            break;
    }
    return hipGetLastError();
}

}  // namespace pm
