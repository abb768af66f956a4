// param_amd/csrc/embbag_fwd_persist.hip -- persistent form of the batched EmbeddingBag(sum) forward for large fixed-pooling
// requests (round 5).  Same reference call sites as embbag_fwd.hip (train/compute/pt/pytorch_emb.py:40,61,
// train/comms/pt/pytorch_dist_backend.py:221,845-849), same arithmetic (adds in index order per lane: bit-identical to the
// sequential fp32 sum and to embbag_fwd_kernel); what changes is WHO touches the output and WHEN.
//
// embbag_fwd_kernel ends every 32-bag tile with a workgroup-wide barrier, a 16 KB burst of stores issued by the pooling waves
// themselves, the workgroup's exit and a new workgroup's two dependent round trips (offsets, then indices) before its first row
// load: 4.7 % of the bytes cost 10-12 % of the launch.  Here:
//   * the grid is (workgroups per CU) x 256 CUs; workgroup b walks the virtual block ids b, b + grid, b + 2 grid, ... of the
//     SAME block -> (table, tile) map (common.h: block_to_tile), so the XCD / table locality of the dispatch order is kept: in
//     iteration k the workgroups of one XCD hold consecutive tiles of one table;
//   * NPW pooling waves + ONE helper wave per workgroup.  The helper stages tile k + A's offsets and indices into an LDS ring
//     (its own loads, its own vmcnt) and drains tile k's pooled rows from LDS to HBM with 16-byte non-temporal stores; the pooling
//     waves issue row loads and LDS accesses only -- no store ever sits in their vmcnt queue, and nothing waits for a barrier:
//     the hand-offs are monotonic LDS words (`staged`, `done[slot]`), polled with s_sleep.  A wave that finishes its bags of
//     tile k starts tile k + 1 at once;
//   * after the first round the drains of different workgroups are spread over time instead of reaching the memory
//     controllers together at the end of a dispatch round.
// Progress: the helper stages tile j only after draining tile j - nslot, which needs the pooling waves to finish that tile, which
// needs it staged -- true by induction from the prologue; no workgroup waits for another workgroup, so a grid larger than
// the chip's residency merely queues.
#include <type_traits>

#include "common.h"
#include "fwd_elem.h"

namespace pm {
namespace {

using namespace fwd;

constexpr int kPsMaxSlots = 8;

struct PsCtl {                       // LDS control block of a workgroup (zeroed before the one block-wide barrier)
    uint32_t staged;                 // tiles staged so far (helper -> pooling waves), monotonic
    uint32_t done[kPsMaxSlots];      // per ring slot: bag slots pooled so far (pooling waves -> helper), monotonic over the slot's uses
    uint32_t meta[kPsMaxSlots];      // per ring slot: 1 = the tile's indices are in LDS
    uint32_t pad[32 - 1 - 2 * kPsMaxSlots];
};
static_assert(sizeof(PsCtl) == 128, "control block is 128 bytes");

__host__ __device__ inline size_t ps_off_bytes(int tb) { return (static_cast<size_t>(tb + 2) / 2 * 2) * sizeof(int64_t); }
__host__ __device__ inline size_t ps_slot_bytes(int tb, int cap, bool weighted, int row_floats) {
    return ps_off_bytes(tb) + static_cast<size_t>(cap) * 4 * (weighted ? 2 : 1) + static_cast<size_t>(tb) * row_floats * sizeof(float);
}

__device__ __forceinline__ uint32_t lds_poll(const uint32_t* w) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// Every spin is bounded: a hand-off that never arrives (a bug, by the progress argument above) aborts the launch -- the caller
// gets a HIP error at its next synchronisation -- instead of hanging the device.  2^25 polls of >= 128 cycles: seconds.
constexpr uint32_t kPsSpinCap = 1u << 25;
__device__ __forceinline__ void wait_at_least(const uint32_t* w, uint32_t v, int nap) {
    for (uint32_t n = 0; lds_poll(w) < v; ++n) {
        if (n >= kPsSpinCap) __builtin_trap();
        if (nap == 1) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(2);
    }
}

template <typename WT, int G, int UNROLL, bool WEIGHTED, int NPW>
__global__ void __launch_bounds__((NPW + 1) * kWave) __attribute__((amdgpu_num_sgpr(72))) embbag_fwd_persist_kernel(const KParams p) {
    constexpr int VEC = Elem<WT>::kVec;
    constexpr int GPW = kWave / G;           // lane groups per wave
    constexpr int NG = NPW * GPW;            // bags pooled concurrently per workgroup
    constexpr int ES = 16 / VEC;             // bytes per table element
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int TB = p.bags_per_block;         // bags per tile = NG * (bags per lane group and tile)
    const int tbg = TB / NG;
    const int nslot = p.ps_slots;
    const int cap = p.idx_cap;
    const size_t off_bytes = ps_off_bytes(TB);
    const size_t slot_bytes = ps_slot_bytes(TB, cap, WEIGHTED, p.stage_out);
    PsCtl* ctl = reinterpret_cast<PsCtl*>(smem);
    char* ring = smem + sizeof(PsCtl);
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave);
    const int lane = static_cast<int>(threadIdx.x) % kWave;
    const int total = p.T * p.tiles_per_table;
    const int bid = static_cast<int>(blockIdx.x), stride = static_cast<int>(gridDim.x);
    const int n_it = bid < total ? (total - 1 - bid) / stride + 1 : 0;

    if (threadIdx.x < sizeof(PsCtl) / 4) reinterpret_cast<uint32_t*>(ctl)[threadIdx.x] = 0;
    __syncthreads();                         // the only block-wide barrier of the kernel

    auto tile_of = [&](int k, int& t, int64_t& bag0, int& nb) {
        int tile;
        block_to_tile(p, t, tile, bid + k * stride);
        bag0 = p.bag_begin + static_cast<int64_t>(tile) * TB;
        const int64_t left = p.bag_begin + p.bag_count - bag0;
        nb = left < TB ? static_cast<int>(left) : TB;
    };

    if (wave == NPW) {
        // ------------------------------------------------ helper wave: stage ahead, drain behind ------------------------------
        auto stage = [&](int k) {
            int t, nb;
            int64_t bag0;
            tile_of(k, t, bag0, nb);
            const int slot = k % nslot;
            char* sl = ring + slot * slot_bytes;
            int64_t* s_off = reinterpret_cast<int64_t*>(sl);
            int32_t* s_idx = reinterpret_cast<int32_t*>(sl + off_bytes);
            float* s_w = reinterpret_cast<float*>(s_idx + cap);
            const int64_t g0 = static_cast<int64_t>(t) * p.B + bag0;
            const int64_t off = bag_start_or_end(p, g0 + (lane <= nb ? lane : nb));      // TB + 1 <= 64 lanes (host)
            if (lane <= nb) s_off[lane] = off;
            const int64_t base = __shfl(off, 0, kWave);
            const int64_t cnt = __shfl(off, nb, kWave) - base;
            const bool ok = cnt <= cap;
            if (ok) {
                const int n = static_cast<int>(cnt);
                auto copy = [&](auto wide_c) {                           // one instantiation per index type: no per-load branch, so
                    constexpr bool WIDE = decltype(wide_c)::value;         // the four loads of a trip are in flight together
                    for (int i0 = 0; i0 < n; i0 += 4 * kWave) {
                        int32_t v[4];
                        float w[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = i0 + u * kWave + lane;
                            const int64_t j = base + (i < n ? i : n - 1);
                            v[u] = WIDE ? static_cast<int32_t>(as_global<int64_t>(p.indices)[j]) : as_global<int32_t>(p.indices)[j];
                            if (WEIGHTED) w[u] = as_global<float>(p.psw)[j];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = i0 + u * kWave + lane;
                            if (i < n) {
                                s_idx[i] = v[u];
                                if (WEIGHTED) s_w[i] = w[u];
                            }
                        }
                    }
                };
                if (p.idx64) copy(std::true_type{}); else copy(std::false_type{});
            }
            asm volatile("" ::: "memory");                            // the tile's LDS stores stay ahead of its publication
            if (lane == 0) {
                ctl->meta[slot] = ok ? 1u : 0u;
                asm volatile("" ::: "memory");
                __hip_atomic_store(&ctl->staged, static_cast<uint32_t>(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        auto drain = [&](int k) {
            const int slot = k % nslot;
            const uint32_t target = static_cast<uint32_t>(k / nslot + 1) * static_cast<uint32_t>(TB);
            wait_at_least(&ctl->done[slot], target, 2);
            asm volatile("" ::: "memory");
            int t, nb;
            int64_t bag0;
            tile_of(k, t, bag0, nb);
            const int D = p.dims[t];
            const float* s_out = reinterpret_cast<const float*>(ring + slot * slot_bytes + off_bytes + static_cast<size_t>(cap) * 4 * (WEIGHTED ? 2 : 1));
            float* out_t = p.io + p.out_offsets[t] + bag0 * p.out_stride;
            const int q = D / 4;                                      // 16-byte pieces per row
            const int n = nb * q;
            auto piece = [&](int bg, int c4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(s_out + static_cast<size_t>(bg) * D + c4 * 4);
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out_t + static_cast<int64_t>(bg) * p.out_stride + c4 * 4));
            };
            if ((q & (q - 1)) == 0) {
                const int lg = 31 - __builtin_clz(static_cast<unsigned>(q));
                for (int i = lane; i < n; i += kWave) piece(i >> lg, i & (q - 1));
            } else {
                for (int i = lane; i < n; i += kWave) piece(i / q, i % q);
            }
            asm volatile("" ::: "memory");                            // the slot's LDS reads stay ahead of its next staging
        };
        const int ahead = nslot - 1;
        for (int k = 0; k < ahead && k < n_it; ++k) stage(k);
        for (int k = 0; k < n_it; ++k) {
            if (k + ahead < n_it) stage(k + ahead);                   // into the slot tile k - 1 left (drained in the last trip)
            drain(k);
        }
        return;
    }

    // ---------------------------------------------------- pooling waves: row loads and LDS only -------------------------------
    const int gid = wave * GPW + lane / G;   // lane group of the workgroup
    const int lig = lane % G;
    const bool nt = p.nt_loads != 0;
    for (int k = 0; k < n_it; ++k) {
        const int slot = k % nslot;
        wait_at_least(&ctl->staged, static_cast<uint32_t>(k) + 1, 1);
        asm volatile("" ::: "memory");
        int t, nb;
        int64_t bag0;
        tile_of(k, t, bag0, nb);
        char* sl = ring + slot * slot_bytes;
        const int64_t* s_off = reinterpret_cast<const int64_t*>(sl);
        const int32_t* s_idx = reinterpret_cast<const int32_t*>(sl + off_bytes);
        const float* s_w = reinterpret_cast<const float*>(s_idx + cap);
        float* s_out = reinterpret_cast<float*>(sl + off_bytes + static_cast<size_t>(cap) * 4 * (WEIGHTED ? 2 : 1));
        const bool staged = __builtin_amdgcn_readfirstlane(ctl->meta[slot]) != 0;
        const int D = p.dims[t];
        const int64_t row_bytes = static_cast<int64_t>(D) * ES;
        const char* W = reinterpret_cast<const char*>(p.tables[t]);
        const int64_t base = s_off[0];

        for (int rep = 0; rep < tbg; ++rep) {
            const int bg = gid + rep * NG;
            if (bg >= nb) continue;
            const int64_t s = s_off[bg];
            const int64_t e = s_off[bg + 1];
            auto pool = [&](auto staged_c) {
                constexpr bool ST = decltype(staged_c)::value;
                for (int c = lig * VEC; c < D; c += G * VEC) {
                    const char* Wc = W + static_cast<int64_t>(c) * ES;
                    float acc[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
                    int64_t j = s;
                    for (; j + UNROLL <= e; j += UNROLL) {               // full batches: UNROLL row loads in flight, then ordered adds
                        u32x4 raw[UNROLL];
                        float w[UNROLL];
                        int64_t r[UNROLL];
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u) {
                            const int64_t jj = j + u;
                            r[u] = ST ? static_cast<int64_t>(s_idx[jj - base]) : load_index(p.indices, jj, p.idx64);
                            if (WEIGHTED) w[u] = ST ? s_w[jj - base] : as_global<float>(p.psw)[jj];
                        }
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u) raw[u] = load16(Wc + row_offset<ST>(r[u], row_bytes), nt);
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u) {
                            float f[VEC];
                            Elem<WT>::widen(raw[u], f);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) acc[i] = WEIGHTED ? fmaf(w[u], f[i], acc[i]) : acc[i] + f[i];
                        }
                    }
                    if (j < e) {                                          // tail (< UNROLL lookups): straight-line, same order of adds
                        u32x4 raw[UNROLL];
                        float w[UNROLL];
                        int64_t r[UNROLL];
#pragma unroll
                        for (int u = 0; u < UNROLL - 1; ++u) {
                            const int64_t jj = j + u < e ? j + u : e - 1;
                            r[u] = ST ? static_cast<int64_t>(s_idx[jj - base]) : load_index(p.indices, jj, p.idx64);
                            if (WEIGHTED) w[u] = ST ? s_w[jj - base] : as_global<float>(p.psw)[jj];
                        }
#pragma unroll
                        for (int u = 0; u < UNROLL - 1; ++u) raw[u] = load16(Wc + row_offset<ST>(r[u], row_bytes), nt);
#pragma unroll
                        for (int u = 0; u < UNROLL - 1; ++u) {
                            if (j + u < e) {
                                float f[VEC];
                                Elem<WT>::widen(raw[u], f);
#pragma unroll
                                for (int i = 0; i < VEC; ++i) acc[i] = WEIGHTED ? fmaf(w[u], f[i], acc[i]) : acc[i] + f[i];
                            }
                        }
                    }
                    f32x4* o4 = reinterpret_cast<f32x4*>(s_out + static_cast<size_t>(bg) * D + c);
#pragma unroll
                    for (int i = 0; i < VEC; i += 4) o4[i / 4] = f32x4{acc[i], acc[i + 1], acc[i + 2], acc[i + 3]};
                }
            };
            if (staged) pool(std::true_type{}); else pool(std::false_type{});
        }
        asm volatile("" ::: "memory");                                    // the pooled rows' LDS stores stay ahead of the arrival
        if (lane == 0) __hip_atomic_fetch_add(&ctl->done[slot], static_cast<uint32_t>(GPW * tbg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

int g_cus = 0;            // compute units of the device (queried once)
int device_cus() {
    if (g_cus <= 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) g_cus = n;
        else g_cus = 256;
    }
    return g_cus;
}

template <typename WT, int G, int UNROLL, int NPW>
hipError_t launch_n(const KParams& p, int wgs_per_cu, hipStream_t stream) {
    const bool weighted = p.psw != nullptr;
    const size_t lds = sizeof(PsCtl) + static_cast<size_t>(p.ps_slots) * ps_slot_bytes(p.bags_per_block, p.idx_cap, weighted, p.stage_out);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int64_t total = static_cast<int64_t>(p.T) * p.tiles_per_table;
    int64_t grid = static_cast<int64_t>(device_cus()) * wgs_per_cu;
    if (grid > total) grid = total;
    if (grid >= kXcds) grid = grid / kXcds * kXcds;       // virtual block ids keep their XCD (b + k * grid = b mod 8)
    auto kern = weighted ? embbag_fwd_persist_kernel<WT, G, UNROLL, true, NPW> : embbag_fwd_persist_kernel<WT, G, UNROLL, false, NPW>;
    static thread_local const void* attr_set[2] = {nullptr, nullptr};
    if (lds > 64 * 1024 && attr_set[weighted ? 1 : 0] != reinterpret_cast<const void*>(kern)) {
        const hipError_t h = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (h != hipSuccess) return h;
        attr_set[weighted ? 1 : 0] = reinterpret_cast<const void*>(kern);
    }
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3((NPW + 1) * kWave), lds, stream, p);
    return hipGetLastError();
}

template <typename WT, int G>
hipError_t launch_u(const KParams& p, int unroll, int pool_waves, int wgs_per_cu, hipStream_t stream) {
#define PM_PS(U_) (pool_waves == 7 ? launch_n<WT, G, U_, 7>(p, wgs_per_cu, stream) : launch_n<WT, G, U_, 4>(p, wgs_per_cu, stream))
    switch (unroll) {
        case 1: case 2: return PM_PS(2);
        default: return PM_PS(4);
    }
#undef PM_PS
}

template <typename WT>
hipError_t launch_g(const KParams& p, int max_dim, int unroll, int pool_waves, int wgs_per_cu, hipStream_t stream) {
    switch (group_lanes(max_dim, Elem<WT>::kVec)) {
        case 8: return launch_u<WT, 8>(p, unroll, pool_waves, wgs_per_cu, stream);
        case 16: return launch_u<WT, 16>(p, unroll, pool_waves, wgs_per_cu, stream);
        case 32: return launch_u<WT, 32>(p, unroll, pool_waves, wgs_per_cu, stream);
        default: return launch_u<WT, 64>(p, unroll, pool_waves, wgs_per_cu, stream);
    }
}

}  // namespace

size_t fwd_persist_lds_bytes(int tile_bags, int idx_cap, bool weighted, int row_floats, int slots) {
    return sizeof(PsCtl) + static_cast<size_t>(slots) * ps_slot_bytes(tile_bags, idx_cap, weighted, row_floats);
}

hipError_t launch_embbag_fwd_persist(const KParams& p, int weight_dtype, int max_dim, int unroll, int pool_waves, int wgs_per_cu,
                                     hipStream_t stream) {
    switch (weight_dtype) {
        case PM_F32: return launch_g<float>(p, max_dim, unroll, pool_waves, wgs_per_cu, stream);
        case PM_BF16: return launch_g<bf16_t>(p, max_dim, unroll, pool_waves, wgs_per_cu, stream);
        default: return launch_g<f16_t>(p, max_dim, unroll, pool_waves, wgs_per_cu, stream);
    }
}

}  // namespace pm
