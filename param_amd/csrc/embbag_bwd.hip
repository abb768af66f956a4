// param_amd/csrc/embbag_bwd.hip -- EmbeddingBag backward: scatter-add into the table(s).
//
//     dst_t[indices[j], :] += alpha * psw[j] * grad(t, bag(j))[:]
//
// Replaces aten::_embedding_bag_dense_backward (autograd of the module built at
// train/compute/pt/pytorch_emb.py:179) when dst is a dense fp32 gradient buffer with
// alpha = 1, and the fused fbgemm TBE backward+update reached at
// train/comms/pt/pytorch_dist_backend.py:854-857 /
// split_table_batched_embeddings_ops.py:318-324 when dst is the table and alpha = -lr.
//
// Same tiling as the forward (one workgroup = a tile of bags of one table, offsets and
// indices staged in LDS, a group of G lanes owns a bag).  Each lane loads its 16-byte
// (fp32 dst) / 32-byte (16-bit dst) column slice of the bag's gradient row ONCE into
// registers and then, per lookup, issues hardware float atomics on the destination row:
//   fp32 dst : 4 x global_atomic_add_f32 per lane  (consecutive lanes -> consecutive dwords)
//   bf16/f16 : 4 x global_atomic_pk_add_{bf16,f16} per lane (two elements per atomic)
// The atomics are fire-and-forget (no return value), performed at the L2, so HBM sees one
// read-modify-write of each touched line.  Duplicate rows (Zipf heads) are handled by the
// atomics; the accumulation order is not fixed (parity: 1e-5 relative, tests/).
#include "common.h"

namespace pm {
namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

struct DstF32 {
    static constexpr int kVec = 4;  // elements per lane
    static constexpr int kES = 4;
    __device__ static __forceinline__ void add(char* row, const float (&v)[4]) {
        PM_GLOBAL float* q = as_global<float>(row);   // global_atomic_add_f32 (no return), not flat_atomic
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_amdgcn_global_atomic_fadd_f32(q + k, v[k]);
    }
};
struct DstBF16 {
    static constexpr int kVec = 8;
    static constexpr int kES = 2;
    __device__ static __forceinline__ void add(char* row, const float (&v)[8]) {
        PM_GLOBAL bf16x2* q = (PM_GLOBAL bf16x2*)(row);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bf16x2 x;
            x[0] = static_cast<__bf16>(v[2 * k]);
            x[1] = static_cast<__bf16>(v[2 * k + 1]);
            __builtin_amdgcn_global_atomic_fadd_v2bf16(q + k, x);
        }
    }
};
struct DstF16 {
    static constexpr int kVec = 8;
    static constexpr int kES = 2;
    __device__ static __forceinline__ void add(char* row, const float (&v)[8]) {
        PM_GLOBAL f16x2* q = (PM_GLOBAL f16x2*)(row);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f16x2 x;
            x[0] = static_cast<_Float16>(v[2 * k]);
            x[1] = static_cast<_Float16>(v[2 * k + 1]);
            __builtin_amdgcn_global_atomic_fadd_v2f16(q + k, x);
        }
    }
};

template <typename DST, int G, bool WEIGHTED>
__global__ void __launch_bounds__(kBlock) embbag_bwd_kernel(const KParams p) {
    constexpr int VEC = DST::kVec;
    constexpr int NG = kBlock / G;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    int t, tile;
    block_to_tile(p, t, tile);
    if (t >= p.T) return;

    int nb;
    int64_t* s_off;
    int32_t* s_idx;
    float* s_w;
    const bool staged = stage_tile<WEIGHTED>(p, t, tile, smem, nb, s_off, s_idx, s_w);
    const int64_t base = s_off[0];

    const int gid = threadIdx.x / G;
    const int lig = threadIdx.x % G;
    const int D = p.dims[t];
    const int64_t row_bytes = static_cast<int64_t>(D) * DST::kES;
    char* W = reinterpret_cast<char*>(const_cast<void*>(p.tables[t]));
    const int64_t bag0 = p.bag_begin + static_cast<int64_t>(tile) * p.bags_per_block;
    const float* grad_t = p.io + p.out_offsets[t];

    for (int bg = gid; bg < nb; bg += NG) {
        const int64_t s = s_off[bg];
        const int64_t e = s_off[bg + 1];
        if (s == e) continue;
        const float* grow = grad_t + grad_bag_offset(bag0 + bg, p.out_stride, p.gblk_shift, p.gblk_extra);

        for (int c = lig * VEC; c < D; c += G * VEC) {
            float g[VEC];
            const f32x4* g4 = reinterpret_cast<const f32x4*>(grow + c);
#pragma unroll
            for (int k = 0; k < VEC; k += 4) {
                const f32x4 v = __builtin_nontemporal_load(g4 + k / 4);
                g[k] = v.x; g[k + 1] = v.y; g[k + 2] = v.z; g[k + 3] = v.w;
            }
            if (!WEIGHTED) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) g[k] *= p.alpha;
            }
            char* Wc = W + static_cast<int64_t>(c) * DST::kES;
            for (int64_t j = s; j < e; ++j) {
                const int64_t r = staged ? static_cast<int64_t>(s_idx[j - base])
                                         : load_index(p.indices, j, p.idx64);
                if (WEIGHTED) {
                    const float sc = p.alpha * (staged ? s_w[j - base] : as_global<float>(p.psw)[j]);
                    float gw[VEC];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) gw[k] = sc * g[k];
                    DST::add(Wc + r * row_bytes, gw);
                } else {
                    DST::add(Wc + r * row_bytes, g);
                }
            }
        }
    }
}

template <typename DST, int G>
hipError_t launch_w(const KParams& p, hipStream_t stream) {
    const bool weighted = p.psw != nullptr;
    const int grid = p.T * p.tiles_per_table;
    const size_t lds = tile_lds_bytes(p.bags_per_block, p.idx_cap, weighted);
    if (weighted)
        hipLaunchKernelGGL((embbag_bwd_kernel<DST, G, true>), dim3(grid), dim3(kBlock), lds, stream, p);
    else
        hipLaunchKernelGGL((embbag_bwd_kernel<DST, G, false>), dim3(grid), dim3(kBlock), lds, stream, p);
    return hipGetLastError();
}

template <typename DST>
hipError_t launch_g(const KParams& p, int max_dim, hipStream_t stream) {
    switch (group_lanes(max_dim, DST::kVec)) {
        case 8: return launch_w<DST, 8>(p, stream);
        case 16: return launch_w<DST, 16>(p, stream);
        case 32: return launch_w<DST, 32>(p, stream);
        default: return launch_w<DST, 64>(p, stream);
    }
}

}  // namespace

hipError_t launch_embbag_bwd(const KParams& p, int dst_dtype, int max_dim, hipStream_t stream) {
    switch (dst_dtype) {
        case PM_F32: return launch_g<DstF32>(p, max_dim, stream);
        case PM_BF16: return launch_g<DstBF16>(p, max_dim, stream);
        default: return launch_g<DstF16>(p, max_dim, stream);
    }
}

}  // namespace pm
