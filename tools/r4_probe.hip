// tools/r4_probe.hip -- round 4, visit 1: what do the building blocks of a "rows looked up once go bag-major" backward cost on
// MI355X?  Standalone (no torch): hipcc --offload-arch=gfx950 -O3 -o build/r4_probe tools/r4_probe.hip ; run on the GPU box.
//   mark_global  : per lookup one returning atomicOr on a per-table `seen` bitmap (+ one on `dup` for repeats); agent scope and
//                  workgroup scope (the latter is only correct when a table's lookups all run on one XCD: timing only)
//   mark_lds     : workgroup (table, slice) scans the table's lookups 3x: LDS bitmaps seen / dup of its slice of the (hashed) row
//                  space, count, ordered emit of the lookups of duplicated rows + the dup bitmap in global memory
//   rmw_unique   : bag-major read-modify-write of rows whose dup bit is clear, the bag's gradient slice in registers
// Prints one JSON object per measurement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define PM_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// indices: uniform (alpha == 0) or a bounded-Pareto stand-in for Zipf(alpha) ranks (hot rows = low ids)
__global__ void gen_indices(int64_t* idx, int64_t n, int64_t rows, double alpha, uint64_t seed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(seed ^ mix64((uint64_t)i));
        int64_t r;
        if (alpha == 0.0) {
            r = (int64_t)(h % (uint64_t)rows);
        } else {
            const double u = ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            const double e = 1.0 - alpha;                          // negative
            const double x = pow(1.0 - u * (1.0 - pow((double)rows, e)), 1.0 / e);
            r = (int64_t)x - 1;
            if (r < 0) r = 0;
            if (r >= rows) r = rows - 1;
        }
        idx[i] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// mark through global atomics
template <int SCOPE>
__global__ void __launch_bounds__(256) mark_global(const int64_t* idx, int64_t per_table, int T, uint32_t* seen, uint32_t* dup,
                                                   uint32_t words_per_table, uint32_t hmask, int xcd_affine) {
    const int tiles = (int)((per_table + 4095) / 4096);
    int t, tile;
    if (xcd_affine) {
        const int x = blockIdx.x % 8, slot = blockIdx.x / 8;
        t = x + 8 * (slot / tiles);
        tile = slot % tiles;
    } else {
        t = blockIdx.x / tiles;
        tile = blockIdx.x % tiles;
    }
    if (t >= T) return;
    const int64_t base = (int64_t)t * per_table + (int64_t)tile * 4096;
    const int cnt = (int)((per_table - (int64_t)tile * 4096) < 4096 ? (per_table - (int64_t)tile * 4096) : 4096);
    uint32_t* sw = seen + (size_t)t * words_per_table;
    uint32_t* dw = dup + (size_t)t * words_per_table;
    int64_t r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int q = k * 256 + threadIdx.x;
        r[k] = idx[base + (q < cnt ? q : cnt - 1)];
    }
    uint32_t old[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t h = (uint32_t)r[k] & hmask;
        old[k] = (k * 256 + threadIdx.x < cnt) ? __hip_atomic_fetch_or(sw + (h >> 5), 1u << (h & 31), __ATOMIC_RELAXED, SCOPE) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t h = (uint32_t)r[k] & hmask;
        if ((k * 256 + threadIdx.x < cnt) && (old[k] >> (h & 31)) & 1u) __hip_atomic_fetch_or(dw + (h >> 5), 1u << (h & 31), __ATOMIC_RELAXED, SCOPE);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// mark + emit through LDS slices.  Workgroup (t, s) owns the hashed rows h = row & hmask with (h & (S - 1)) == s; bit b = h >> lgS
// of its 2^19-bit slice.  Global dup bit of a row: word_off[t] + ((s << 19 | b) >> 5).
constexpr int kSliceBits = 1 << 19;
constexpr int kSliceWords = kSliceBits / 32;   // 16384 words = 64 KB
constexpr int kMT = 1024;

struct MarkArgs {
    const int64_t* idx;
    int64_t per_table;
    int T, S, lgS;
    uint32_t hmask;
    uint32_t* dup_out;         // [T][S * kSliceWords]
    uint32_t* cursor;          // [T]
    uint32_t* keys;            // [N] region of table t at t * per_table
    uint32_t* vals;
    int tshift;
    uint32_t pooling;
    int scans;                 // 1: mark only, 2: + count, 3: + emit
    int xcd_affine;
};

__global__ void __launch_bounds__(kMT) mark_lds(const MarkArgs a) {
    extern __shared__ uint32_t lds[];
    uint32_t* seen = lds;
    uint32_t* dup = lds + kSliceWords;
    __shared__ uint32_t s_w[kMT / 64];
    __shared__ uint32_t s_base;
    int t, s;
    if (a.xcd_affine) {
        const int x = blockIdx.x % 8, slot = blockIdx.x / 8;
        t = x + 8 * (slot / a.S);
        s = slot % a.S;
    } else {
        t = blockIdx.x / a.S;
        s = blockIdx.x % a.S;
    }
    if (t >= a.T) return;
    for (int i = threadIdx.x; i < 2 * kSliceWords; i += kMT) lds[i] = 0u;
    __syncthreads();
    const int64_t n = a.per_table;
    const int64_t* ip = a.idx + (int64_t)t * n;
    const uint32_t smask = (uint32_t)a.S - 1u;
    // scan 1: mark
    constexpr int U = 8;
    for (int64_t i0 = 0; i0 < n; i0 += (int64_t)kMT * U) {
        int64_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * kMT + threadIdx.x;
            r[u] = ip[i < n ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * kMT + threadIdx.x;
            const uint32_t h = (uint32_t)r[u] & a.hmask;
            if (i < n && (h & smask) == (uint32_t)s) {
                const uint32_t b = h >> a.lgS, w = b >> 5, bit = 1u << (b & 31);
                if (!(dup[w] & bit)) {
                    const uint32_t old = atomicOr(&seen[w], bit);
                    if (old & bit) atomicOr(&dup[w], bit);
                }
            }
        }
    }
    __syncthreads();
    // the slice's dup bits to global memory
    uint32_t* dout = a.dup_out + ((size_t)t * a.S + s) * kSliceWords;
    for (int i = threadIdx.x; i < kSliceWords; i += kMT) dout[i] = dup[i];
    if (a.scans < 2) return;
    // scan 2: lookups of duplicated rows in this slice
    uint32_t mine = 0;
    for (int64_t i0 = 0; i0 < n; i0 += (int64_t)kMT * U) {
        int64_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * kMT + threadIdx.x;
            r[u] = ip[i < n ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * kMT + threadIdx.x;
            const uint32_t h = (uint32_t)r[u] & a.hmask;
            const uint32_t b = h >> a.lgS;
            if (i < n && (h & smask) == (uint32_t)s && ((dup[b >> 5] >> (b & 31)) & 1u)) ++mine;
        }
    }
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
    uint32_t tot = mine;
    for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
    if (lane == 0) s_w[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t all = 0;
        for (int w = 0; w < kMT / 64; ++w) all += s_w[w];
        s_base = atomicAdd(&a.cursor[t], all);
    }
    __syncthreads();
    if (a.scans < 3) return;
    // scan 3: ordered emit -- thread holds 4 consecutive positions of a chunk of 4096
    uint32_t run = s_base;
    uint32_t* kout = a.keys + (size_t)t * n;
    uint32_t* vout = a.vals + (size_t)t * n;
    for (int64_t c0 = 0; c0 < n; c0 += 4 * kMT) {
        const int64_t p0 = c0 + 4 * (int64_t)threadIdx.x;
        int64_t r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = ip[p0 + u < n ? p0 + u : n - 1];
        uint32_t f = 0, cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t h = (uint32_t)r[u] & a.hmask;
            const uint32_t b = h >> a.lgS;
            const bool hit = p0 + u < n && (h & smask) == (uint32_t)s && ((dup[b >> 5] >> (b & 31)) & 1u);
            f |= (hit ? 1u : 0u) << u;
            cnt += hit ? 1u : 0u;
        }
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        __syncthreads();                     // s_w free (previous chunk's readers done)
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, all = 0;
        for (int w = 0; w < kMT / 64; ++w) {
            const uint32_t x = s_w[w];
            if (w < wave) wbase += x;
            all += x;
        }
        uint32_t o = run + wbase + incl - cnt;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if ((f >> u) & 1u) {
                kout[o] = ((uint32_t)t << a.tshift) | (uint32_t)r[u];
                vout[o] = (uint32_t)((p0 + u) / a.pooling);
                ++o;
            }
        }
        run += all;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bag-major read-modify-write of the rows whose dup bit is clear
struct RmwArgs {
    void* const* tables;
    const int64_t* idx;
    const float* grad;          // [B, T * D]
    const uint32_t* dup;        // [T][dup_words]
    uint32_t dup_words, hmask, smask;
    int lgS, sshift;
    int64_t B;
    int T, L, D;
    float alpha;
    int xcd_affine;
    int bags_per_block;
};

__device__ __forceinline__ u32x4 ld16(const char* p) { return *(const PM_GLOBAL u32x4*)(p); }
__device__ __forceinline__ void st16_sc1(char* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"((PM_GLOBAL u32x4*)p), "v"(v) : "memory");
}

template <int ES, int G, int UNROLL>
__global__ void __launch_bounds__(256) rmw_unique(const RmwArgs a) {
    constexpr int VEC = 16 / ES;            // elements per lane: 4 (fp32) or 8 (bf16)
    constexpr int NG = 256 / G;
    extern __shared__ uint32_t s_idx[];     // bags_per_block * L entries: row | dup << 31
    const int tiles = (int)((a.B + a.bags_per_block - 1) / a.bags_per_block);
    int t, tile;
    if (a.xcd_affine) {
        const int x = blockIdx.x % 8, slot = blockIdx.x / 8;
        t = x + 8 * (slot / tiles);
        tile = slot % tiles;
    } else {
        t = blockIdx.x / tiles;
        tile = blockIdx.x % tiles;
    }
    if (t >= a.T) return;
    const int64_t bag0 = (int64_t)tile * a.bags_per_block;
    const int nb = (int)((a.B - bag0) < a.bags_per_block ? (a.B - bag0) : a.bags_per_block);
    const int cnt = nb * a.L;
    const int64_t base = ((int64_t)t * a.B + bag0) * a.L;
    const uint32_t* dw = a.dup + (size_t)t * a.dup_words;
    // stage indices + dup flags: all index loads, then all bitmap loads
    constexpr int SU = 4;
    for (int i0 = 0; i0 < cnt; i0 += 256 * SU) {
        int64_t r[SU];
        uint32_t w[SU], g[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * 256 + threadIdx.x;
            r[u] = a.idx[base + (i < cnt ? i : cnt - 1)];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const uint32_t h = (uint32_t)r[u] & a.hmask;
            g[u] = ((h & a.smask) << a.sshift) | (h >> a.lgS);
            w[u] = dw[g[u] >> 5];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = i0 + u * 256 + threadIdx.x;
            if (i < cnt) s_idx[i] = (uint32_t)r[u] | (((w[u] >> (g[u] & 31)) & 1u) << 31);
        }
    }
    __syncthreads();
    const int gid = threadIdx.x / G, lig = threadIdx.x % G;
    char* W = (char*)a.tables[t];
    const int64_t row_bytes = (int64_t)a.D * ES;
    const int c = lig * VEC;
    if (c >= a.D) return;
    for (int bg = gid; bg < nb; bg += NG) {
        const float* gp = a.grad + (bag0 + bg) * (int64_t)a.T * a.D + (int64_t)t * a.D + c;
        float ga[VEC];
#pragma unroll
        for (int k = 0; k < VEC; k += 4) {
            const f32x4 x = *(const PM_GLOBAL f32x4*)(gp + k);
            ga[k] = a.alpha * x.x; ga[k + 1] = a.alpha * x.y; ga[k + 2] = a.alpha * x.z; ga[k + 3] = a.alpha * x.w;
        }
        const uint32_t* si = s_idx + bg * a.L;
        for (int j = 0; j < a.L; j += UNROLL) {
            u32x4 raw[UNROLL];
            uint32_t rr[UNROLL];
            char* ptr[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) rr[u] = si[j + u < a.L ? j + u : a.L - 1];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const bool live = j + u < a.L && !(rr[u] >> 31);
                ptr[u] = live ? W + (uint64_t)(rr[u] & 0x7fffffffu) * (uint32_t)row_bytes + c * ES : (char*)gp;
                raw[u] = ld16(ptr[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const bool live = j + u < a.L && !(rr[u] >> 31);
                if (live) {
                    u32x4 o;
                    if (ES == 4) {
                        o.x = __float_as_uint(__uint_as_float(raw[u].x) + ga[0]);
                        o.y = __float_as_uint(__uint_as_float(raw[u].y) + ga[1]);
                        o.z = __float_as_uint(__uint_as_float(raw[u].z) + ga[2]);
                        o.w = __float_as_uint(__uint_as_float(raw[u].w) + ga[3]);
                    } else {
                        const uint32_t wv[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
                        uint32_t ov[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float lo = __uint_as_float(wv[q] << 16) + ga[(2 * q) % VEC];
                            const float hi = __uint_as_float(wv[q] & 0xffff0000u) + ga[(2 * q + 1) % VEC];
                            const uint32_t ul = __float_as_uint(lo), uh = __float_as_uint(hi);
                            ov[q] = ((ul + 0x7fffu + ((ul >> 16) & 1u)) >> 16) | ((uh + 0x7fffu + ((uh >> 16) & 1u)) & 0xffff0000u);
                        }
                        o.x = ov[0]; o.y = ov[1]; o.z = ov[2]; o.w = ov[3];
                    }
                    st16_sc1(ptr[u], o);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) asm volatile("" : : "v"(raw[u]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename F>
static double time_ms(F&& fn, int iters, int warm = 3) {
    for (int i = 0; i < warm; ++i) fn();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int T32 = 48, T16 = 64, D = 128, L = 20;
    const int64_t B = 8192, R = 10000000;
    const int64_t per_table = B * L;
    const bool skip_rmw = argc > 1 && atoi(argv[1]) == 1;
    const int Tmax = T16;
    const int64_t N = per_table * Tmax;
    int64_t *idx_u, *idx_z;
    CK(hipMalloc(&idx_u, N * 8));
    CK(hipMalloc(&idx_z, N * 8));
    for (int t = 0; t < Tmax; ++t) {
        hipLaunchKernelGGL(gen_indices, dim3(640), dim3(256), 0, 0, idx_u + t * per_table, per_table, R, 0.0, 1000 + t);
        hipLaunchKernelGGL(gen_indices, dim3(640), dim3(256), 0, 0, idx_z + t * per_table, per_table, R, 1.05, 2000 + t);
    }
    CK(hipDeviceSynchronize());
    {   // how skewed is the stand-in?  distinct rows of table 0 on the host
        std::vector<int64_t> h(per_table);
        CK(hipMemcpy(h.data(), idx_z, per_table * 8, hipMemcpyDeviceToHost));
        std::vector<uint8_t> seen(R, 0);
        int64_t distinct = 0, once = 0;
        for (auto r : h) { if (seen[r] < 2) ++seen[r]; }
        int64_t singles = 0;
        for (auto r : h) { if (seen[r] == 1) ++singles; }
        for (int64_t r = 0; r < R; ++r) distinct += seen[r] ? 1 : 0;
        (void)once;
        printf("{\"what\": \"zipf stand-in, table 0\", \"lookups\": %lld, \"distinct_rows\": %lld, \"lookups_of_rows_seen_once\": %lld}\n",
               (long long)per_table, (long long)distinct, (long long)singles);
    }

    // ---- global-atomic mark: exact bitmaps (2^24 bits >= 10 M rows) and hashed 2^21
    for (int hb : {24, 21}) {
        const uint32_t words = (1u << hb) / 32, hmask = (1u << hb) - 1u;
        uint32_t *seen, *dup;
        CK(hipMalloc(&seen, (size_t)T32 * words * 4));
        CK(hipMalloc(&dup, (size_t)T32 * words * 4));
        const double ms_set = time_ms([&] { CK(hipMemsetAsync(seen, 0, (size_t)T32 * words * 4, 0)); CK(hipMemsetAsync(dup, 0, (size_t)T32 * words * 4, 0)); }, 20);
        printf("{\"what\": \"memset seen+dup\", \"hash_bits\": %d, \"MB\": %.1f, \"ms\": %.4f}\n", hb, 2.0 * T32 * words * 4 / 1e6, ms_set);
        const int tiles = (int)((per_table + 4095) / 4096);
        for (int z = 0; z < 2; ++z) {
            const int64_t* ix = z ? idx_z : idx_u;
            for (int xa = 0; xa < 2; ++xa) {
                const double a = time_ms([&] {
                    CK(hipMemsetAsync(seen, 0, (size_t)T32 * words * 4, 0));
                    CK(hipMemsetAsync(dup, 0, (size_t)T32 * words * 4, 0));
                    hipLaunchKernelGGL((mark_global<__HIP_MEMORY_SCOPE_AGENT>), dim3(T32 * tiles), dim3(256), 0, 0, ix, per_table, T32, seen, dup, words, hmask, xa);
                }, 10);
                const double w = time_ms([&] {
                    CK(hipMemsetAsync(seen, 0, (size_t)T32 * words * 4, 0));
                    CK(hipMemsetAsync(dup, 0, (size_t)T32 * words * 4, 0));
                    hipLaunchKernelGGL((mark_global<__HIP_MEMORY_SCOPE_WORKGROUP>), dim3(T32 * tiles), dim3(256), 0, 0, ix, per_table, T32, seen, dup, words, hmask, xa);
                }, 10);
                printf("{\"what\": \"mark_global (memset included)\", \"hash_bits\": %d, \"indices\": \"%s\", \"xcd_affine\": %d, \"agent_ms\": %.4f, \"workgroup_ms\": %.4f, \"memset_ms\": %.4f}\n",
                       hb, z ? "zipf" : "uniform", xa, a, w, ms_set);
                fflush(stdout);
            }
        }
        CK(hipFree(seen));
        CK(hipFree(dup));
    }

    // ---- LDS mark / count / emit, S slices of 2^19 bits per table
    {
        CK(hipFuncSetAttribute((const void*)mark_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kSliceWords * 4));
        uint32_t *dup_out, *cursor, *keys, *vals;
        const int Smax = 8;
        CK(hipMalloc(&dup_out, (size_t)Tmax * Smax * kSliceWords * 4));
        CK(hipMalloc(&cursor, Tmax * 4));
        CK(hipMalloc(&keys, N * 4));
        CK(hipMalloc(&vals, N * 4));
        for (int T : {T32, T16}) {
            for (int S : {4, 8}) {
                if ((size_t)T * S > 512) continue;
                int lgS = 0;
                while ((1 << lgS) < S) ++lgS;
                for (int z = 0; z < 2; ++z) {
                    for (int scans = 1; scans <= 3; ++scans) {
                        MarkArgs a;
                        a.idx = z ? idx_z : idx_u;
                        a.per_table = per_table;
                        a.T = T; a.S = S; a.lgS = lgS;
                        a.hmask = (uint32_t)S * kSliceBits - 1u;
                        a.dup_out = dup_out; a.cursor = cursor; a.keys = keys; a.vals = vals;
                        a.tshift = 24; a.pooling = L; a.scans = scans; a.xcd_affine = 1;
                        const double ms = time_ms([&] {
                            CK(hipMemsetAsync(cursor, 0, Tmax * 4, 0));
                            hipLaunchKernelGGL(mark_lds, dim3(T * S), dim3(kMT), 2 * kSliceWords * 4, 0, a);
                        }, 10);
                        CK(hipGetLastError());
                        std::vector<uint32_t> cur(T);
                        CK(hipMemcpy(cur.data(), cursor, T * 4, hipMemcpyDeviceToHost));
                        uint64_t tot = 0;
                        for (auto c : cur) tot += c;
                        printf("{\"what\": \"mark_lds\", \"tables\": %d, \"slices\": %d, \"hash_bits\": %d, \"indices\": \"%s\", \"scans\": %d, \"ms\": %.4f, \"dup_lookups\": %llu, \"dup_frac\": %.4f}\n",
                               T, S, 19 + lgS, z ? "zipf" : "uniform", scans, ms, (unsigned long long)tot, (double)tot / ((double)T * per_table));
                        fflush(stdout);
                    }
                }
            }
        }
        if (skip_rmw) return 0;

        // ---- rmw_unique with a real dup bitmap (from mark_lds, S = 4) and with an empty one
        std::vector<void*> tabs_h(Tmax);
        void** tabs_d;
        CK(hipMalloc(&tabs_d, Tmax * sizeof(void*)));
        float* grad;
        CK(hipMalloc(&grad, (size_t)B * Tmax * D * 4));
        CK(hipMemset(grad, 0, (size_t)B * Tmax * D * 4));
        for (int pass = 0; pass < 2; ++pass) {
            const int ES = pass == 0 ? 4 : 2;
            const int T = pass == 0 ? T32 : T16;
            for (int t = 0; t < T; ++t) CK(hipMalloc(&tabs_h[t], (size_t)R * D * ES));
            for (int t = 0; t < T; ++t) CK(hipMemsetAsync(tabs_h[t], 0, (size_t)R * D * ES, 0));
            CK(hipMemcpy(tabs_d, tabs_h.data(), T * sizeof(void*), hipMemcpyHostToDevice));
            for (int z = 0; z < 2; ++z) {
                for (int real = 0; real < 2; ++real) {
                    const int S = 4, lgS = 2;
                    MarkArgs m;
                    m.idx = z ? idx_z : idx_u;
                    m.per_table = per_table;
                    m.T = T; m.S = S; m.lgS = lgS;
                    m.hmask = (uint32_t)S * kSliceBits - 1u;
                    m.dup_out = dup_out; m.cursor = cursor; m.keys = keys; m.vals = vals;
                    m.tshift = 24; m.pooling = L; m.scans = 1; m.xcd_affine = 1;
                    if (real) hipLaunchKernelGGL(mark_lds, dim3(T * S), dim3(kMT), 2 * kSliceWords * 4, 0, m);
                    else CK(hipMemsetAsync(dup_out, 0, (size_t)T * S * kSliceWords * 4, 0));
                    RmwArgs a;
                    a.tables = tabs_d; a.idx = m.idx; a.grad = grad; a.dup = dup_out;
                    a.dup_words = S * kSliceWords; a.hmask = m.hmask; a.smask = S - 1; a.lgS = lgS; a.sshift = 19;
                    a.B = B; a.T = T; a.L = L; a.D = D; a.alpha = -1e-6f; a.xcd_affine = 1; a.bags_per_block = 32;
                    const int tiles = (int)((B + a.bags_per_block - 1) / a.bags_per_block);
                    const size_t lds = (size_t)a.bags_per_block * L * 4;
                    double ms2, ms4, ms8 = 0;
                    if (ES == 4) {
                        ms2 = time_ms([&] { hipLaunchKernelGGL((rmw_unique<4, 32, 2>), dim3(T * tiles), dim3(256), lds, 0, a); }, 10);
                        ms4 = time_ms([&] { hipLaunchKernelGGL((rmw_unique<4, 32, 4>), dim3(T * tiles), dim3(256), lds, 0, a); }, 10);
                        ms8 = time_ms([&] { hipLaunchKernelGGL((rmw_unique<4, 32, 8>), dim3(T * tiles), dim3(256), lds, 0, a); }, 10);
                    } else {
                        ms2 = time_ms([&] { hipLaunchKernelGGL((rmw_unique<2, 16, 2>), dim3(T * tiles), dim3(256), lds, 0, a); }, 10);
                        ms4 = time_ms([&] { hipLaunchKernelGGL((rmw_unique<2, 16, 4>), dim3(T * tiles), dim3(256), lds, 0, a); }, 10);
                        ms8 = time_ms([&] { hipLaunchKernelGGL((rmw_unique<2, 16, 8>), dim3(T * tiles), dim3(256), lds, 0, a); }, 10);
                    }
                    CK(hipGetLastError());
                    printf("{\"what\": \"rmw_unique\", \"elem_bytes\": %d, \"tables\": %d, \"indices\": \"%s\", \"dup_bitmap\": \"%s\", \"unroll2_ms\": %.4f, \"unroll4_ms\": %.4f, \"unroll8_ms\": %.4f}\n",
                           ES, T, z ? "zipf" : "uniform", real ? "marked" : "empty (every lookup applied)", ms2, ms4, ms8);
                    fflush(stdout);
                }
            }
            for (int t = 0; t < T; ++t) CK(hipFree(tabs_h[t]));
        }
    }
    return 0;
}
