// param_amd/csrc/seg_sort.hip -- the key sort of the sorted EmbeddingBag backward, round 3: per-TABLE segments whose
// bounds, pooling factors and key widths are established ON THE DEVICE from the request itself, so that every request --
// fixed pooling, per-table multi-hot pooling (Criteo), ragged bags, batch slices, per-sample weights -- gets
//   * the table bits out of the sort (a table's lookups are a contiguous range of the table-major request),
//   * the first pass formed straight from the index array wherever a table's bags all have the same length (verified by
//     reading the offsets, not taken on the caller's word), and
//   * compact output (no padding keys: a batch slice is just a shorter segment).
// Reference semantics served: the sort inside aten::_embedding_bag_dense_backward / fbgemm's TBE backward (call sites
// train/comms/pt/pytorch_dist_backend.py:854-857, train/compute/python/workloads/pytorch/
// split_table_batched_embeddings_ops.py:318-324); request layout :93-135,191-208.
//
// Kernels (gfx950, no inter-workgroup communication inside a kernel -- kernel boundaries are the only synchronisation, as
// in radix_sort.hip):
//   seg_prep_tables_kernel   one workgroup per table: segment start / count from the offsets, pooling factor if every bag
//                            of the (sliced) table has the same length, key bits of the table from rows[t]
//   seg_prep_scan_kernel     workgroup 0: output start and first tile of every segment, one 32-byte descriptor per tile; the
//                            other workgroups, only for tables WITHOUT a pooling factor (ragged, weighted): (key, bag) per
//                            lookup at request positions (binary search over LDS-staged offsets) -- they exit at once otherwise
//   seg_hist / seg_scan / seg_scatter   one radix pass over 4096-element tiles: per-tile digit counts, per-segment
//                            exclusive prefix (+ absolute bucket starts), stable scatter
//   MODE 1 / 2 (one global partition pass, then buckets):
//   seg_local_kernel         every (table, digit) bucket of up to 4096 pairs is sorted by its remaining key bits inside LDS:
//                            up to 1024 pairs by ONE WAVE (four buckets per workgroup, no workgroup barrier), up to 4096 by
//                            the workgroup; larger buckets were put on a list by the scan kernel ...
//   seg_l2_prep_kernel + the pass kernels again   ... and become the segments of a second-level LSD sort over their
//                            remaining bits (persistent grids: a request without such buckets pays a few empty launches)
// MODE 0 runs ceil(rbits / 8) global LSD passes (ascending (table, row, position) order, like round 2's segmented sort);
// MODE 1 partitions on the LOW row digit (order (table, row & 255, row >> 8, position): buckets balanced under any skew, but
//        neighbours in the sorted array are not neighbours in the table -- the apply kernel pays for that under skew);
// MODE 2 partitions on the TOP row digit (ascending order; a skewed head makes its bucket a second-level segment).
// Equal keys end up adjacent and in request order in every mode -- all the apply kernel needs.
#include <cstdlib>

#include "common.h"

namespace pm {
namespace {

constexpr int kT = 256;                 // threads per workgroup
constexpr int kWaves = kT / kWave;      // 4
constexpr int kTile = 4096;             // elements per radix tile (16 per thread)
constexpr int kTileItems = kTile / kT;  // 16
constexpr int kRadix = 256;
constexpr uint32_t kWaveCap = 1024;     // bucket-local sort by one wave: 16 pairs per lane
constexpr uint32_t kLocalCap = kTile;   // ... by one workgroup; larger buckets go to the second level
constexpr int kL2Grid = 512;            // persistent grid of the second-level passes
constexpr int kBuildBags = 1024;        // bags per workgroup of the key-building kernel

struct SegHeader {
    uint32_t n_total;    // pairs in all segments (= length of the sorted arrays)
    uint32_t n_tiles;    // radix tiles of the first level
    uint32_t n_l2;       // buckets on the second-level list (= second-level segments)
    uint32_t n_tiles2;   // radix tiles of the second level
    uint32_t pad[12];
};

// one per radix tile: a pass kernel's workgroup learns everything about its tile from one 32-byte load
struct TileDesc {
    uint32_t seg;        // segment (level 1: table; level 2: index into the second-level list)
    uint32_t cnt;        // elements of the tile
    uint32_t in_base;    // level 1: request position of the tile's first element
    uint32_t out_base;   // position of the tile's first element in the sorted arrays
    uint32_t first;      // position of the tile's first element inside its segment
    uint32_t pooling;    // level 1: the segment's pooling factor (0: keys were built)
    uint32_t rbits;      // bits to sort in this segment | (first bit << 8)
    uint32_t magic;      // floor(2^32 / pooling) + 1: x / pooling = mulhi(x, magic), one step too high at most (fast_div)
};

// x / d for a d whose magic = floor(2^32 / d) + 1 is at hand: the multiply-high estimate is the quotient or one more
__device__ __forceinline__ uint32_t fast_div(uint32_t x, uint32_t d, uint32_t magic) {
    uint32_t q = d == 1 ? x : __umulhi(x, magic);
    if (static_cast<uint64_t>(q) * d > x) --q;
    return q;
}

__device__ __forceinline__ int bits_for_dev(uint64_t n_values) {   // bits needed to represent 0 .. n_values - 1
    return n_values <= 1 ? 0 : 64 - __builtin_clzll(n_values - 1);
}

// digit of pass `pass` for a segment that sorts `bits` bits starting at bit `bit0`: (shift, mask).  mode 0 / 1: LSD, 8 bits per
// pass from bit0 (mode 1 runs pass 0 only).  mode 2: pass 0 = the top digit.  A width of 0 makes the pass a stable copy.
__device__ __forceinline__ void pass_digit(int mode, int pass, uint32_t rbits_packed, int& shift, uint32_t& mask) {
    const int bits = static_cast<int>(rbits_packed & 255u), bit0 = static_cast<int>(rbits_packed >> 8);
    int w;
    if (mode == 2) {
        w = bits < 8 ? bits : 8;
        shift = bit0 + bits - w;
    } else {
        shift = bit0 + 8 * pass;
        w = bits - 8 * pass;
        w = w < 0 ? 0 : (w > 8 ? 8 : w);
        if (w == 0) shift = 0;
    }
    mask = (1u << w) - 1u;
}
// bits a (table, digit) bucket still has to order after the partition pass: [lo, hi)
__device__ __forceinline__ void local_bits(int mode, int rbits, int& lo, int& hi) {
    if (mode == 2) { lo = 0; hi = rbits - 8; }
    else { lo = 8; hi = rbits; }
    if (hi < lo) hi = lo;
}

__device__ __forceinline__ uint64_t match_digit8(uint32_t d, bool valid) {
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// exclusive scan of one value per thread over the 256 threads of the workgroup; s_tmp: kWaves words
__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t* s_tmp) {
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += up;
    }
    if (lane == kWave - 1) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += s_tmp[w];
    __syncthreads();
    return base + incl - v;
}

// the lanes of a wave hand LDS data to each other: LDS operations of one wave execute in order, the compiler must keep them so
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------------------
// prep 1: one workgroup (1024 threads) per table
__global__ void __launch_bounds__(1024) seg_prep_tables_kernel(const void* offsets, int idx64, const int64_t* rows, int T, int64_t B,
                                                               int64_t N, int64_t bag_begin, int64_t bag_count, int force_ragged,
                                                               SegDesc* desc) {
    const int t = blockIdx.x;
    const int64_t TB = static_cast<int64_t>(T) * B;
    const int64_t g0 = static_cast<int64_t>(t) * B + bag_begin;
    auto off_at = [&](int64_t g) -> int64_t { return g < TB ? load_index(offsets, g, idx64) : N; };
    const int64_t s = off_at(g0);
    const int64_t e = off_at(g0 + bag_count);
    const int64_t cnt = e > s ? e - s : 0;
    const int64_t L = (bag_count > 0 && cnt > 0 && cnt % bag_count == 0) ? cnt / bag_count : 0;
    int bad = (L == 0 || force_ragged) ? 1 : 0;
    if (!bad) {
        // every bag of the (sliced) table must start where a pooling factor of L puts it; 8 independent loads per round trip
        for (int64_t i0 = threadIdx.x; i0 < bag_count; i0 += 8 * 1024) {
            int64_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + u * 1024;
                v[u] = i < bag_count ? off_at(g0 + i) : s + i * L;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] != s + (i0 + u * 1024) * L) bad = 1;
        }
    }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        SegDesc d;
        d.in_start = static_cast<uint32_t>(s);
        d.count = static_cast<uint32_t>(cnt);
        d.out_start = 0;
        d.pooling = bad ? 0u : static_cast<uint32_t>(L);
        d.tile_base = 0;
        d.ntiles = static_cast<uint32_t>((cnt + kTile - 1) / kTile);
        d.rbits = static_cast<uint32_t>(bits_for_dev(static_cast<uint64_t>(rows[t])));
        d.pad = 0;
        desc[t] = d;
    }
}

// inclusive scan of two values per thread over 1024 threads in LDS (Hillis-Steele: one launch per sort, nothing cleverer needed)
__device__ __forceinline__ void scan2_1024(uint32_t* s_a, uint32_t* s_b) {
    const int t = threadIdx.x;
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t a = t >= off ? s_a[t - off] : 0u;
        const uint32_t b = t >= off ? s_b[t - off] : 0u;
        __syncthreads();
        s_a[t] += a;
        s_b[t] += b;
        __syncthreads();
    }
}

// keys / values at request positions for the tables that have no pooling factor (and, WEIGHTED, for all: the value is the
// lookup's position, its bag goes to bag_of): workgroup (chunk of 1024 bags, table) of the preparation kernel below.
template <typename K, bool WEIGHTED>
__device__ __forceinline__ void build_keys_chunk(int t, int chunk, const void* indices, const void* offsets, int idx64, int T, int64_t B,
                                                 int64_t N, int64_t bag_begin, int64_t bag_count, const SegDesc* desc, int tshift, K* keys,
                                                 uint32_t* vals, uint32_t* bag_of, int64_t* s_off) {
    constexpr int kBags = kBuildBags;
    if (!WEIGHTED && desc[t].pooling > 0) return;
    const int64_t bag0 = bag_begin + static_cast<int64_t>(chunk) * kBags;
    const int64_t left = bag_begin + bag_count - bag0;
    if (left <= 0) return;
    const int nb = left < kBags ? static_cast<int>(left) : kBags;
    const int64_t TB = static_cast<int64_t>(T) * B;
    const int64_t g0 = static_cast<int64_t>(t) * B + bag0;
    for (int i = threadIdx.x; i <= nb; i += blockDim.x) s_off[i] = (g0 + i < TB) ? load_index(offsets, g0 + i, idx64) : N;
    __syncthreads();
    const int64_t base = s_off[0], end = s_off[nb];
    for (int64_t j = base + threadIdx.x; j < end; j += blockDim.x) {
        int lo = 0, hi = nb;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= j) lo = mid; else hi = mid;
        }
        const uint32_t bag = static_cast<uint32_t>(bag0 + lo);
        keys[j] = (static_cast<K>(t) << tshift) | static_cast<K>(load_index(indices, j, idx64));
        if (WEIGHTED) {
            vals[j] = static_cast<uint32_t>(j);
            bag_of[j] = bag;
        } else {
            vals[j] = bag;
        }
    }
}

// prep 2, workgroup 0 (1024 threads, T <= 1024): exclusive scans over the tables, tile descriptors, header.  The OTHER
// workgroups of the launch build the (key, bag) pairs of the tables without a pooling factor (one workgroup per 1024 bags of a
// table; they exit at once for tables that have one) -- the two jobs need only prep 1's results, and a launch saved is ~6 us.
template <typename K, bool WEIGHTED>
__global__ void __launch_bounds__(1024) seg_prep_scan_kernel(SegDesc* desc, int T, SegHeader* hdr, TileDesc* tiles, uint32_t tiles_cap,
                                                             const void* indices, const void* offsets, int idx64, int64_t B, int64_t N,
                                                             int64_t bag_begin, int64_t bag_count, int tshift, K* keys, uint32_t* vals,
                                                             uint32_t* bag_of, int chunks_per_table) {
    __shared__ uint32_t s_cnt[1024], s_til[1024], s_tb[1025], s_out[1024];
    __shared__ int64_t s_off[kBuildBags + 1];
    if (blockIdx.x > 0) {
        const int w = static_cast<int>(blockIdx.x) - 1;
        build_keys_chunk<K, WEIGHTED>(w / chunks_per_table, w % chunks_per_table, indices, offsets, idx64, T, B, N, bag_begin, bag_count,
                                      desc, tshift, keys, vals, bag_of, s_off);
        return;
    }
    const int t = threadIdx.x;
    const uint32_t c = t < T ? desc[t].count : 0u;
    const uint32_t nt = t < T ? desc[t].ntiles : 0u;
    s_cnt[t] = c;
    s_til[t] = nt;
    __syncthreads();
    scan2_1024(s_cnt, s_til);
    const uint32_t out_start = s_cnt[t] - c, tile_base = s_til[t] - nt;
    const uint32_t n_total = s_cnt[1023], n_tiles = s_til[1023];
    __syncthreads();
    if (t < T) {
        desc[t].out_start = out_start;
        desc[t].tile_base = tile_base;
        s_tb[t] = tile_base;
        s_out[t] = out_start;
    }
    if (t == 0) {
        s_tb[T] = n_tiles;
        hdr->n_total = n_total;
        hdr->n_tiles = n_tiles;
        hdr->n_l2 = 0;
        hdr->n_tiles2 = 0;
    }
    __syncthreads();
    const uint32_t n_write = n_tiles < tiles_cap ? n_tiles : tiles_cap;
    for (uint32_t g = t; g < n_write; g += 1024) {
        int lo = 0, hi = T;         // largest t with s_tb[t] <= g (segments without tiles repeat their neighbour's base)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_tb[mid] <= g) lo = mid; else hi = mid;
        }
        const SegDesc d = desc[lo];   // fields of prep 1 only: this kernel's own writes to other entries may not be visible yet
        const uint32_t first = (g - s_tb[lo]) * static_cast<uint32_t>(kTile);
        TileDesc td;
        td.seg = static_cast<uint32_t>(lo);
        td.cnt = (d.count - first) < static_cast<uint32_t>(kTile) ? d.count - first : static_cast<uint32_t>(kTile);
        td.in_base = d.in_start + first;
        td.out_base = s_out[lo] + first;
        td.first = first;
        td.pooling = d.pooling;
        td.rbits = d.rbits;
        td.magic = d.pooling > 1 ? static_cast<uint32_t>(0x100000000ull / d.pooling) + 1u : 0u;
        tiles[g] = td;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// What a pass reads: pass 0 of level 1 the request (index array, or the built keys of tables without a pooling factor) at
// request positions; every other pass the previous pass's output at output positions.
template <typename K>
struct PassSrc {
    const void* indices;    // level 1, pass 0
    int idx64;
    const K* keys;          // level 1, pass 0: built keys (request positions); otherwise: the previous output
    const uint32_t* vals;
    int first;              // 1: level 1, pass 0
    int tshift;
    uint32_t bag_begin;
};

// tiles are taken g = blockIdx.x, + gridDim.x, ... < *n_tiles: level 1 launches one workgroup per possible tile, level 2 a
// persistent grid (the tile count of level 2 is known only on the device, and is zero for most requests)
template <typename K>
__global__ void __launch_bounds__(kT) seg_hist_kernel(const TileDesc* tiles, const uint32_t* n_tiles, const PassSrc<K> src, int mode, int pass,
                                                      uint32_t* bh) {
    __shared__ uint32_t h[kRadix];
    // the first descriptor is fetched together with the tile count, not after it (grids never exceed the descriptor arrays):
    // a workgroup's start-up is a chain of dependent loads, and every link costs a memory latency that nothing hides
    TileDesc td = tiles[blockIdx.x];
    const uint32_t nt = *n_tiles;
    for (uint32_t g = blockIdx.x; g < nt; g += gridDim.x) {
        if (g != blockIdx.x) td = tiles[g];
        const uint32_t cnt = td.cnt;
        int shift;
        uint32_t mask;
        pass_digit(mode, pass, td.rbits, shift, mask);
        h[threadIdx.x] = 0;
        __syncthreads();
        const int lane = threadIdx.x % kWave;
        const uint64_t base = src.first ? td.in_base : td.out_base;
#pragma unroll 4
        for (int k = 0; k < kTileItems; ++k) {
            const uint32_t i = static_cast<uint32_t>(k) * kT + threadIdx.x;
            const bool valid = i < cnt;
            uint32_t dg = 0u;
            if (valid) {
                // the row digit of pass 0 can always be taken from the index array (built keys carry the same row bits)
                const uint64_t row = src.first ? static_cast<uint64_t>(load_index(src.indices, static_cast<int64_t>(base + i), src.idx64))
                                               : static_cast<uint64_t>(src.keys[base + i]);
                dg = static_cast<uint32_t>(row >> shift) & mask;
            }
            // a wave whose keys share the digit adds once (top digits of a skewed head, small tables); else one LDS atomic per lane
            const uint64_t vmask = __ballot(valid);
            const uint32_t firstd = __builtin_amdgcn_readfirstlane(dg);
            const bool uniform = __ballot(valid && dg != firstd) == 0 && (vmask & 1ull);
            if (uniform) {
                if (lane == 0) atomicAdd(&h[firstd], static_cast<uint32_t>(__popcll(vmask)));
            } else if (valid) {
                atomicAdd(&h[dg], 1u);
            }
        }
        __syncthreads();
        bh[static_cast<uint64_t>(g) * kRadix + threadIdx.x] = h[threadIdx.x];
        __syncthreads();
    }
}

// per segment: exclusive prefix of the tile counts per digit (in place), absolute start and size of every (segment, digit)
// bucket; classify != 0 (level 1 of modes 1 / 2): buckets too large for the local sort go to the second-level list.
// 1024 threads = 4 row chunks x 256 digits, 16 independent loads per round trip: a segment of 200 tiles (the 100-hot Criteo
// table) is 4 round trips per phase instead of the 25 a 256-thread walk needs -- that walk was the long pole of every pass of
// the Criteo sort (one workgroup per segment, and one segment holds half the lookups).
constexpr int kScanChunks = 4;
constexpr int kScanU = 16;
__global__ void __launch_bounds__(kRadix * kScanChunks) seg_scan_kernel(SegHeader* hdr, const SegDesc* desc, const uint32_t* n_seg_dev,
                                                                        uint32_t n_seg_host, uint32_t* bh, uint32_t* bstart, uint32_t* bcnt,
                                                                        int classify, int mode, uint32_t* l2_list) {
    __shared__ uint32_t s_sum[kScanChunks][kRadix];
    __shared__ uint32_t s_tmp[kWaves];
    const int d = threadIdx.x % kRadix;
    const int c = threadIdx.x / kRadix;
    const uint32_t n_seg = n_seg_dev ? *n_seg_dev : n_seg_host;
    for (uint32_t t = blockIdx.x; t < n_seg; t += gridDim.x) {
        const SegDesc sd = desc[t];
        const uint64_t r0 = sd.tile_base;
        const uint32_t per = (sd.ntiles + kScanChunks - 1) / kScanChunks;
        const uint32_t ra = c * per < sd.ntiles ? c * per : sd.ntiles;
        const uint32_t rb = ra + per < sd.ntiles ? ra + per : sd.ntiles;
        uint32_t sum = 0;
        for (uint32_t r = ra; r < rb; r += kScanU) {
            uint32_t v[kScanU];
#pragma unroll
            for (int u = 0; u < kScanU; ++u) v[u] = (r + u < rb) ? bh[(r0 + r + u) * kRadix + d] : 0u;
#pragma unroll
            for (int u = 0; u < kScanU; ++u) sum += v[u];
        }
        s_sum[c][d] = sum;
        __syncthreads();
        uint32_t run = 0, total = 0;
#pragma unroll
        for (int cc = 0; cc < kScanChunks; ++cc) {
            const uint32_t x = s_sum[cc][d];
            if (cc < c) run += x;
            total += x;
        }
        for (uint32_t r = ra; r < rb; r += kScanU) {     // in-place rewrite: the loads of a batch are issued before its stores
            uint32_t v[kScanU];
#pragma unroll
            for (int u = 0; u < kScanU; ++u) v[u] = (r + u < rb) ? bh[(r0 + r + u) * kRadix + d] : 0u;
#pragma unroll
            for (int u = 0; u < kScanU; ++u) {
                if (r + u < rb) bh[(r0 + r + u) * kRadix + d] = run;
                run += v[u];
            }
        }
        // exclusive scan of the digit totals over the 256 digits (the threads of chunk 0; everybody keeps the barriers)
        uint32_t incl = (c == 0) ? total : 0u;
        const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += up;
        }
        if (c == 0 && lane == kWave - 1) s_tmp[wave] = incl;
        __syncthreads();
        if (c == 0) {
            uint32_t base = 0;
            for (int w = 0; w < wave; ++w) base += s_tmp[w];
            const uint32_t start = sd.out_start + base + incl - total;
            const uint32_t b = t * kRadix + d;
            bstart[b] = start;
            bcnt[b] = total;
            if (classify && total > kLocalCap) {
                int lo, hi;
                local_bits(mode, static_cast<int>(sd.rbits), lo, hi);
                if (hi > lo) l2_list[atomicAdd(&hdr->n_l2, 1u)] = b;
            }
        }
        __syncthreads();     // s_sum / s_tmp are rewritten by the next segment of a looping workgroup
    }
}

// Stable placement of a tile's elements by one digit, staged in LDS.  The elements sit in registers: thread (wave, lane)
// holds tile positions wave * chunk + r * 64 + lane, r = 0 .. ITEMS-1 (valid: r * 64 + lane < chunk and position < cnt),
// so waves own consecutive runs of the tile and (wave, r, lane) order = position order.  On return s_key / s_val hold
// the tile reordered by digit (stable), s_dstart[d] = first staged position of digit d.
template <typename K, int ITEMS>
__device__ __forceinline__ void tile_stage_by_digit(const K (&key)[ITEMS], const uint32_t (&val)[ITEMS], uint32_t cnt, uint32_t chunk,
                                                    int shift, uint32_t mask, K* s_key, uint32_t* s_val, uint32_t* s_wcnt,
                                                    uint32_t* s_dstart, uint32_t* s_tmp) {
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    for (int i = threadIdx.x; i < kWaves * kRadix; i += kT) s_wcnt[i] = 0;
    __syncthreads();
    uint32_t rank[ITEMS];
    uint32_t* wcnt = s_wcnt + wave * kRadix;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        rank[r] = 0;
        if (static_cast<uint32_t>(r) * kWave < chunk) {     // wave-uniform
            const uint32_t off = static_cast<uint32_t>(r) * kWave + lane;
            const bool valid = off < chunk && wave * chunk + off < cnt;
            const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & mask;
            const uint64_t m = match_digit8(d, valid);
            const uint32_t below = static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
            const uint32_t base = valid ? wcnt[d] : 0u;
            rank[r] = base + below;
            // the lowest lane of each match set advances the digit's counter; a wave executes its LDS operations in program
            // order, so the next round's reads see it
            if (valid && below == 0) wcnt[d] = base + static_cast<uint32_t>(__popcll(m));
        }
    }
    __syncthreads();
    {
        const int d = threadIdx.x;
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t c = s_wcnt[w * kRadix + d];
            s_wcnt[w * kRadix + d] = acc;     // wave w's elements of digit d start this far into the digit's staged run
            acc += c;
        }
        s_dstart[d] = block_excl_scan256(acc, s_tmp);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if (static_cast<uint32_t>(r) * kWave < chunk) {
            const uint32_t off = static_cast<uint32_t>(r) * kWave + lane;
            if (off < chunk && wave * chunk + off < cnt) {
                const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & mask;
                const uint32_t q = s_dstart[d] + wcnt[d] + rank[r];
                s_key[q] = key[r];
                s_val[q] = val[r];
            }
        }
    }
    __syncthreads();
}

// one tile of a scatter pass (inlined into both kernels below: the LDS arrays keep their address space)
template <typename K>
__device__ __forceinline__ void scatter_tile(const TileDesc td, uint32_t g, const PassSrc<K>& src, int mode, int pass, const uint32_t* prefix,
                                             const uint32_t* bstart, K* kout, uint32_t* vout, K* s_key, uint32_t* s_val, uint32_t* s_wcnt,
                                             uint32_t* s_dstart, uint32_t* s_gbase, uint32_t* s_tmp, uint32_t lane_zero) {
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    constexpr uint32_t chunk = kTile / kWaves;
    const uint32_t t = td.seg, cnt = td.cnt, first = td.first;
    int shift;
    uint32_t mask;
    pass_digit(mode, pass, td.rbits, shift, mask);
    const uint64_t base = (src.first ? td.in_base : td.out_base) + lane_zero;
    const bool from_idx = src.first && td.pooling > 0;    // keys formed from the index array, bag = position / pooling
    s_gbase[threadIdx.x] = bstart[t * kRadix + threadIdx.x] + prefix[static_cast<uint64_t>(g) * kRadix + threadIdx.x];
    K key[kTileItems];
    uint32_t val[kTileItems];
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const uint32_t pos = wave * chunk + r * kWave + lane;
        const bool valid = pos < cnt;
        key[r] = 0;
        val[r] = 0;
        if (valid) {
            if (from_idx) {
                key[r] = (static_cast<K>(t) << src.tshift) | static_cast<K>(load_index(src.indices, static_cast<int64_t>(base + pos), src.idx64));
                val[r] = src.bag_begin + fast_div(first + pos, td.pooling, td.magic);
            } else {
                key[r] = src.keys[base + pos];
                val[r] = src.vals[base + pos];
            }
        }
    }
    tile_stage_by_digit<K, kTileItems>(key, val, cnt, chunk, shift, mask, s_key, s_val, s_wcnt, s_dstart, s_tmp);
#pragma unroll 4
    for (int k = 0; k < kTileItems; ++k) {
        const uint32_t q = k * kT + threadIdx.x;
        if (q < cnt) {
            const K kk = s_key[q];
            const uint32_t dg = static_cast<uint32_t>(kk >> shift) & mask;
            const uint64_t o = static_cast<uint64_t>(s_gbase[dg]) + (q - s_dstart[dg]);
            kout[o] = kk;
            vout[o] = s_val[q];
        }
    }
}

// level 1: one workgroup per possible tile (114 VGPRs: the four workgroups per CU that the 38 KB of LDS allow)
template <typename K>
__global__ void __launch_bounds__(kT) seg_scatter_kernel(const TileDesc* tiles, const uint32_t* n_tiles, const PassSrc<K> src, int mode, int pass,
                                                         const uint32_t* prefix, const uint32_t* bstart, K* kout, uint32_t* vout) {
    __shared__ K s_key[kTile];
    __shared__ uint32_t s_val[kTile];
    __shared__ uint32_t s_wcnt[kWaves * kRadix];
    __shared__ uint32_t s_dstart[kRadix];
    __shared__ uint32_t s_gbase[kRadix];
    __shared__ uint32_t s_tmp[kWaves];
    const TileDesc td = tiles[blockIdx.x];      // together with the tile count (see seg_hist_kernel)
    if (blockIdx.x >= *n_tiles) return;
    scatter_tile<K>(td, blockIdx.x, src, mode, pass, prefix, bstart, kout, vout, s_key, s_val, s_wcnt, s_dstart, s_gbase, s_tmp, 0u);
}

// level 2: persistent grid.  (A plain loop around the tile body lets the compiler hoist every lane-dependent address out of
// it -- 179 VGPRs, two workgroups per CU --; the body's addresses therefore hang on a zero the compiler cannot see through.)
template <typename K>
__global__ void __launch_bounds__(kT) seg_scatter_loop_kernel(const TileDesc* tiles, const uint32_t* n_tiles, const PassSrc<K> src, int mode,
                                                              int pass, const uint32_t* prefix, const uint32_t* bstart, K* kout, uint32_t* vout) {
    __shared__ K s_key[kTile];
    __shared__ uint32_t s_val[kTile];
    __shared__ uint32_t s_wcnt[kWaves * kRadix];
    __shared__ uint32_t s_dstart[kRadix];
    __shared__ uint32_t s_gbase[kRadix];
    __shared__ uint32_t s_tmp[kWaves];
    const uint32_t nt = *n_tiles;
#pragma clang loop unroll(disable)
    for (uint32_t g = blockIdx.x; g < nt; g += gridDim.x) {
        uint32_t lane_zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
        scatter_tile<K>(tiles[g], g, src, mode, pass, prefix, bstart, kout, vout, s_key, s_val, s_wcnt, s_dstart, s_gbase, s_tmp, lane_zero);
        __syncthreads();   // s_gbase / the staged tile are rewritten by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bucket-local sort, workgroup form (in place in the b buffers): the bucket's pairs are loaded into registers, ordered by
// their remaining key bits with stable 8-bit rounds through LDS, and written back as one contiguous run.
template <typename K, int ITEMS>
__device__ __forceinline__ void local_sort_bucket(K* kb, uint32_t* vb, uint32_t start, uint32_t n, int lo, int hi, K* s_key,
                                                  uint32_t* s_val, uint32_t* s_wcnt, uint32_t* s_dstart, uint32_t* s_tmp) {
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    const uint32_t chunk = ((n + kT - 1) / kT) * kWave;      // per wave: a multiple of 64, 4 * chunk >= n
    K key[ITEMS];
    uint32_t val[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t off = static_cast<uint32_t>(r) * kWave + lane;
        const uint32_t pos = wave * chunk + off;
        const bool valid = off < chunk && pos < n;
        key[r] = valid ? kb[start + pos] : static_cast<K>(0);
        val[r] = valid ? vb[start + pos] : 0u;
    }
    const int rounds = (hi - lo + 7) / 8;
    for (int rd = 0; rd < rounds; ++rd) {
        const int shift = lo + 8 * rd;
        const int w = (hi - shift) < 8 ? (hi - shift) : 8;
        tile_stage_by_digit<K, ITEMS>(key, val, n, chunk, shift, (1u << w) - 1u, s_key, s_val, s_wcnt, s_dstart, s_tmp);
        if (rd + 1 < rounds) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const uint32_t off = static_cast<uint32_t>(r) * kWave + lane;
                const uint32_t pos = wave * chunk + off;
                if (off < chunk && pos < n) {
                    key[r] = s_key[pos];
                    val[r] = s_val[pos];
                }
            }
            __syncthreads();     // everybody has read the staged tile before the next round overwrites it
        }
    }
    for (uint32_t q = threadIdx.x; q < n; q += kT) {
        kb[start + q] = s_key[q];
        vb[start + q] = s_val[q];
    }
    __syncthreads();             // the LDS tile is reused by the next bucket
}

// ... wave form: one wave sorts one bucket of up to 1024 pairs with its own slice of the LDS arrays and no workgroup barrier
// (the four waves of a workgroup work on four different buckets: a 640-pair bucket -- the uniform benchmark's -- is a few
// wave-steps of work, and four workgroup barriers per round cost more than the work)
template <typename K>
__device__ __forceinline__ void wave_sort_bucket(K* kb, uint32_t* vb, uint32_t start, uint32_t n, int lo, int hi, K* w_key, uint32_t* w_val,
                                                 uint32_t* wcnt) {
    constexpr int ITEMS = kWaveCap / kWave;   // 16
    const int lane = threadIdx.x % kWave;
    K key[ITEMS];
    uint32_t val[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t pos = static_cast<uint32_t>(r) * kWave + lane;
        key[r] = pos < n ? kb[start + pos] : static_cast<K>(0);
        val[r] = pos < n ? vb[start + pos] : 0u;
    }
    const int rounds = (hi - lo + 7) / 8;
    for (int rd = 0; rd < rounds; ++rd) {
        const int shift = lo + 8 * rd;
        const int w = (hi - shift) < 8 ? (hi - shift) : 8;
        const uint32_t mask = (1u << w) - 1u;
#pragma unroll
        for (int i = 0; i < kRadix / kWave; ++i) wcnt[i * kWave + lane] = 0;
        wave_lds_fence();
        uint32_t rank[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            rank[r] = 0;
            if (static_cast<uint32_t>(r) * kWave < n) {     // wave-uniform
                const bool valid = static_cast<uint32_t>(r) * kWave + lane < n;
                const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & mask;
                const uint64_t m = match_digit8(d, valid);
                const uint32_t below = static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
                const uint32_t base = valid ? wcnt[d] : 0u;
                rank[r] = base + below;
                if (valid && below == 0) wcnt[d] = base + static_cast<uint32_t>(__popcll(m));
                wave_lds_fence();
            }
        }
        // exclusive scan of the 256 digit counts by the wave: lane l owns digits 4l .. 4l+3
        {
            uint32_t c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = wcnt[4 * lane + i];
            const uint32_t sum = c[0] + c[1] + c[2] + c[3];
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, kWave);
                if (lane >= off) incl += up;
            }
            uint32_t run = incl - sum;
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wcnt[4 * lane + i] = run;
                run += c[i];
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (static_cast<uint32_t>(r) * kWave + lane < n) {
                const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & mask;
                const uint32_t q = wcnt[d] + rank[r];
                w_key[q] = key[r];
                w_val[q] = val[r];
            }
        }
        wave_lds_fence();
        if (rd + 1 < rounds) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const uint32_t pos = static_cast<uint32_t>(r) * kWave + lane;
                if (pos < n) {
                    key[r] = w_key[pos];
                    val[r] = w_val[pos];
                }
            }
            wave_lds_fence();
        }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t pos = static_cast<uint32_t>(r) * kWave + lane;
        if (pos < n) {
            kb[start + pos] = w_key[pos];
            vb[start + pos] = w_val[pos];
        }
    }
}

// one workgroup per FOUR consecutive (table, digit) buckets: buckets of up to 1024 pairs are taken by one wave each, then
// those of up to 4096 by the whole workgroup; larger ones are second-level segments (seg_scan_kernel listed them)
template <typename K>
__global__ void __launch_bounds__(kT) seg_local_kernel(const SegDesc* desc, const uint32_t* bstart, const uint32_t* bcnt, uint32_t n_buckets,
                                                       int mode, K* kb, uint32_t* vb) {
    __shared__ K s_key[kTile];
    __shared__ uint32_t s_val[kTile];
    __shared__ uint32_t s_wcnt[kWaves * kRadix];
    __shared__ uint32_t s_dstart[kRadix];
    __shared__ uint32_t s_tmp[kWaves];
    const int wave = threadIdx.x / kWave;
    const uint32_t b0 = blockIdx.x * kWaves;
    uint32_t n4[kWaves];
    bool any_wg = false;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        n4[w] = (b0 + w < n_buckets) ? bcnt[b0 + w] : 0u;
        any_wg = any_wg || (n4[w] > kWaveCap && n4[w] <= kLocalCap);
    }
    int lo, hi;
    local_bits(mode, static_cast<int>(desc[b0 / kRadix].rbits), lo, hi);     // the four buckets belong to one table (256 % 4 == 0)
    if (hi <= lo) return;
    {
        const uint32_t n = n4[0] * (wave == 0) + n4[1] * (wave == 1) + n4[2] * (wave == 2) + n4[3] * (wave == 3);
        if (n >= 2 && n <= kWaveCap)
            wave_sort_bucket<K>(kb, vb, bstart[b0 + wave], n, lo, hi, s_key + wave * kWaveCap, s_val + wave * kWaveCap, s_wcnt + wave * kRadix);
    }
    if (!any_wg) return;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        if (n4[w] > kWaveCap && n4[w] <= kLocalCap)
            local_sort_bucket<K, kTileItems>(kb, vb, bstart[b0 + w], n4[w], lo, hi, s_key, s_val, s_wcnt, s_dstart, s_tmp);
    }
}

// second level: the listed buckets become segments (in place in the b buffers) of an LSD sort over their remaining bits
__global__ void __launch_bounds__(1024) seg_l2_prep_kernel(SegHeader* hdr, const SegDesc* desc, const uint32_t* bstart, const uint32_t* bcnt,
                                                           const uint32_t* l2_list, int mode, SegDesc* desc2, uint32_t seg2_cap, TileDesc* tiles2,
                                                           uint32_t tiles2_cap) {
    __shared__ uint32_t s_til[1024], s_dummy[1024];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x;
    uint32_t n2 = hdr->n_l2;
    if (n2 > seg2_cap) n2 = seg2_cap;          // cannot happen (every listed bucket holds > 4096 pairs); keeps the writes in bounds
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n2; c0 += 1024) {
        const uint32_t i = c0 + t;
        uint32_t b = 0, cnt = 0, nt = 0;
        if (i < n2) {
            b = l2_list[i];
            cnt = bcnt[b];
            nt = (cnt + kTile - 1) / kTile;
        }
        s_til[t] = nt;
        s_dummy[t] = 0;
        __syncthreads();
        scan2_1024(s_til, s_dummy);
        const uint32_t tile_base = s_carry + s_til[t] - nt;
        if (i < n2) {
            int lo, hi;
            local_bits(mode, static_cast<int>(desc[b / kRadix].rbits), lo, hi);
            const uint32_t start = bstart[b];
            SegDesc d;
            d.in_start = start;
            d.count = cnt;
            d.out_start = start;
            d.pooling = 0;
            d.tile_base = tile_base;
            d.ntiles = nt;
            d.rbits = static_cast<uint32_t>(hi - lo) | (static_cast<uint32_t>(lo) << 8);
            d.pad = b;
            desc2[i] = d;
            for (uint32_t k = 0; k < nt && tile_base + k < tiles2_cap; ++k) {
                TileDesc td;
                td.seg = i;
                td.first = k * static_cast<uint32_t>(kTile);
                td.cnt = (cnt - td.first) < static_cast<uint32_t>(kTile) ? cnt - td.first : static_cast<uint32_t>(kTile);
                td.in_base = td.out_base = start + td.first;
                td.pooling = 0;
                td.rbits = d.rbits;
                td.magic = 0;
                tiles2[tile_base + k] = td;
            }
        }
        __syncthreads();
        if (t == 1023) s_carry += s_til[1023];
        __syncthreads();
    }
    if (t == 0) hdr->n_tiles2 = s_carry < tiles2_cap ? s_carry : tiles2_cap;
}

inline size_t a256(size_t x) { return (x + 255) / 256 * 256; }
inline size_t tiles_max(size_t n, int T) { return n / kTile + static_cast<size_t>(T) + 1; }
inline size_t seg2_max(size_t n) { return n / kTile + 1; }                 // every second-level segment holds more than a tile
inline size_t tiles2_max(size_t n) {
    const size_t t2 = n / kTile + seg2_max(n) + 1;
    return t2 < static_cast<size_t>(kL2Grid) ? static_cast<size_t>(kL2Grid) : t2;   // the persistent grid reads tiles2[blockIdx.x] before it knows the count
}

struct Scratch {
    SegHeader* hdr;
    SegDesc* desc;
    TileDesc* tiles;
    uint32_t* bh;
    uint32_t* bstart;
    uint32_t* bcnt;
    uint32_t* l2_list;
    SegDesc* desc2;
    TileDesc* tiles2;
    uint32_t* bh2;
    uint32_t* bstart2;
    uint32_t* bcnt2;
    size_t total;
};

Scratch scratch_layout(void* base, size_t n, int T) {
    Scratch s;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = p ? p + off : nullptr; off += a256(bytes); return q; };
    const size_t tm = tiles_max(n, T), nb = static_cast<size_t>(T) * kRadix, s2 = seg2_max(n), t2 = tiles2_max(n);
    s.hdr = reinterpret_cast<SegHeader*>(take(sizeof(SegHeader)));
    s.desc = reinterpret_cast<SegDesc*>(take(sizeof(SegDesc) * static_cast<size_t>(T)));
    s.tiles = reinterpret_cast<TileDesc*>(take(sizeof(TileDesc) * tm));
    s.bh = reinterpret_cast<uint32_t*>(take(4 * tm * kRadix));
    s.bstart = reinterpret_cast<uint32_t*>(take(4 * nb));
    s.bcnt = reinterpret_cast<uint32_t*>(take(4 * nb));
    s.l2_list = reinterpret_cast<uint32_t*>(take(4 * nb));
    s.desc2 = reinterpret_cast<SegDesc*>(take(sizeof(SegDesc) * s2));
    s.tiles2 = reinterpret_cast<TileDesc*>(take(sizeof(TileDesc) * t2));
    s.bh2 = reinterpret_cast<uint32_t*>(take(4 * t2 * kRadix));
    s.bstart2 = reinterpret_cast<uint32_t*>(take(4 * s2 * kRadix));
    s.bcnt2 = reinterpret_cast<uint32_t*>(take(4 * s2 * kRadix));
    s.total = off;
    return s;
}

}  // namespace

size_t seg_sort_scratch_bytes(size_t n_max, int T) { return scratch_layout(nullptr, n_max, T).total; }

const SegDesc* seg_sort_desc(const void* scratch, size_t n_max, int T) { return scratch_layout(const_cast<void*>(scratch), n_max, T).desc; }
const uint32_t* seg_sort_count(const void* scratch, size_t n_max, int T) {
    return &scratch_layout(const_cast<void*>(scratch), n_max, T).hdr->n_total;
}

int seg_sort_passes(int mode, int rbits_max) { return mode == 0 ? (rbits_max <= 0 ? 1 : (rbits_max + 7) / 8) : 1; }
bool seg_sort_result_in_b(int mode, int rbits_max) { return seg_sort_passes(mode, rbits_max) % 2 == 1; }

template <typename K>
hipError_t seg_sort_pairs(const SegSortRequest& rq, int mode, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* bag_of,
                          void* scratch, hipStream_t stream) {
    if (rq.N == 0) return hipSuccess;
    if (rq.T < 1 || rq.T > kSegSortMaxTables || rq.N > 0xffffffffLL) return hipErrorInvalidValue;
    const size_t n = static_cast<size_t>(rq.N);
    const Scratch s = scratch_layout(scratch, n, rq.T);
    const unsigned tm = static_cast<unsigned>(tiles_max(n, rq.T));
    hipLaunchKernelGGL(seg_prep_tables_kernel, dim3(rq.T), dim3(1024), 0, stream, rq.offsets, rq.idx64, rq.rows, rq.T, rq.B, rq.N,
                       rq.bag_begin, rq.bag_count, rq.weighted ? 1 : 0, s.desc);
    {
        const int chunks = rq.bag_count > 0 ? static_cast<int>((rq.bag_count + kBuildBags - 1) / kBuildBags) : 0;
        const dim3 gp(1u + static_cast<unsigned>(chunks) * static_cast<unsigned>(rq.T));
        if (rq.weighted)
            hipLaunchKernelGGL((seg_prep_scan_kernel<K, true>), gp, dim3(1024), 0, stream, s.desc, rq.T, s.hdr, s.tiles, tm, rq.indices,
                               rq.offsets, rq.idx64, rq.B, rq.N, rq.bag_begin, rq.bag_count, rq.tshift, keys_a, vals_a, bag_of, chunks);
        else
            hipLaunchKernelGGL((seg_prep_scan_kernel<K, false>), gp, dim3(1024), 0, stream, s.desc, rq.T, s.hdr, s.tiles, tm, rq.indices,
                               rq.offsets, rq.idx64, rq.B, rq.N, rq.bag_begin, rq.bag_count, rq.tshift, keys_a, vals_a, bag_of, chunks);
    }
    // pass 0 reads the request (or the built keys in the a buffers) and writes the b buffers; later passes alternate, so the
    // sorted pairs end in the b buffers iff the pass count is odd (seg_sort_result_in_b).  Modes 1 / 2 run one global pass and
    // then work in place in b; the a buffers are the second level's spare space.
    const int total = seg_sort_passes(mode, rq.rbits_max);
    PassSrc<K> src;
    src.indices = rq.indices;
    src.idx64 = rq.idx64;
    src.tshift = rq.tshift;
    src.bag_begin = static_cast<uint32_t>(rq.bag_begin);
    for (int p = 0; p < total; ++p) {
        src.first = p == 0 ? 1 : 0;
        src.keys = p == 0 ? keys_a : (p % 2 == 1 ? keys_b : keys_a);
        src.vals = p == 0 ? vals_a : (p % 2 == 1 ? vals_b : vals_a);
        K* kout = (p % 2 == 0) ? keys_b : keys_a;
        uint32_t* vout = (p % 2 == 0) ? vals_b : vals_a;
        hipLaunchKernelGGL((seg_hist_kernel<K>), dim3(tm), dim3(kT), 0, stream, s.tiles, &s.hdr->n_tiles, src, mode, p, s.bh);
        hipLaunchKernelGGL(seg_scan_kernel, dim3(rq.T), dim3(kRadix * kScanChunks), 0, stream, s.hdr, s.desc, static_cast<const uint32_t*>(nullptr),
                           static_cast<uint32_t>(rq.T), s.bh, s.bstart, s.bcnt, (mode != 0 && p == 0) ? 1 : 0, mode, s.l2_list);
        hipLaunchKernelGGL((seg_scatter_kernel<K>), dim3(tm), dim3(kT), 0, stream, s.tiles, &s.hdr->n_tiles, src, mode, p, s.bh, s.bstart, kout,
                           vout);
    }
    if (mode != 0 && rq.rbits_max > 8) {
        const uint32_t nb = static_cast<uint32_t>(rq.T) * kRadix;
        hipLaunchKernelGGL((seg_local_kernel<K>), dim3((nb + kWaves - 1) / kWaves), dim3(kT), 0, stream, s.desc, s.bstart, s.bcnt, nb, mode,
                           keys_b, vals_b);
        // second level: LSD over the remaining bits of the listed buckets, b -> a -> b (an odd pass count gets a copy pass)
        hipLaunchKernelGGL(seg_l2_prep_kernel, dim3(1), dim3(1024), 0, stream, s.hdr, s.desc, s.bstart, s.bcnt, s.l2_list, mode, s.desc2,
                           static_cast<uint32_t>(seg2_max(n)), s.tiles2, static_cast<uint32_t>(tiles2_max(n)));
        int p2 = (rq.rbits_max - 8 + 7) / 8;
        if (p2 % 2) ++p2;
        src.first = 0;
        for (int q = 0; q < p2; ++q) {
            src.keys = (q % 2 == 0) ? keys_b : keys_a;
            src.vals = (q % 2 == 0) ? vals_b : vals_a;
            K* kout = (q % 2 == 0) ? keys_a : keys_b;
            uint32_t* vout = (q % 2 == 0) ? vals_a : vals_b;
            hipLaunchKernelGGL((seg_hist_kernel<K>), dim3(kL2Grid), dim3(kT), 0, stream, s.tiles2, &s.hdr->n_tiles2, src, 0, q, s.bh2);
            hipLaunchKernelGGL(seg_scan_kernel, dim3(kL2Grid / 2), dim3(kRadix * kScanChunks), 0, stream, s.hdr, s.desc2, &s.hdr->n_l2, 0u, s.bh2, s.bstart2,
                               s.bcnt2, 0, 0, static_cast<uint32_t*>(nullptr));
            hipLaunchKernelGGL((seg_scatter_loop_kernel<K>), dim3(kL2Grid), dim3(kT), 0, stream, s.tiles2, &s.hdr->n_tiles2, src, 0, q, s.bh2,
                               s.bstart2, kout, vout);
        }
    }
    return hipGetLastError();
}

template hipError_t seg_sort_pairs<uint32_t>(const SegSortRequest&, int, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, void*,
                                             hipStream_t);
template hipError_t seg_sort_pairs<uint64_t>(const SegSortRequest&, int, uint64_t*, uint64_t*, uint32_t*, uint32_t*, uint32_t*, void*,
                                             hipStream_t);

}  // namespace pm
