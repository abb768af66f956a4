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
// Kernels (gfx950):
//   seg_prep_tables_kernel   one workgroup per table: segment start / count from the offsets, pooling factor if every bag
//                            of the (sliced) table has the same length, key bits of the table from rows[t]
//   seg_prep_scan_kernel     workgroup 0: output start and first tile of every segment, one 32-byte descriptor per tile; the
//                            other workgroups, only for tables WITHOUT a pooling factor (ragged, weighted): (key, bag) per
//                            lookup at request positions (binary search over LDS-staged offsets) -- they exit at once otherwise
//   MODE 0 (default), one kernel per pass:
//   seg_hist_all_kernel      digit counts of ALL passes of every 4096-element tile from one read of the request
//   seg_scan_all_kernel      grid tables x passes: the tables' bucket starts of every pass
//   seg_lookback_pass_kernel a pass: count, publish the tile's digit counts, walk back over the predecessors' published
//                            counts (the only inter-workgroup communication in this file: 16-byte sc1 status rows, a tile waits
//                            for tiles of lower index only), stable scatter
//   MODE 3 (the same order, kernel boundaries as the only synchronisation, as in radix_sort.hip; also the second level of 1 / 2):
//   seg_hist / seg_scan / seg_scatter   one radix pass: per-tile digit counts, per-segment exclusive prefix (+ absolute
//                            bucket starts), stable scatter
//   MODE 1 / 2 (one global partition pass, then buckets):
//   seg_local_kernel         every (table, digit) bucket of up to 4096 pairs is sorted by its remaining key bits inside LDS:
//                            up to 1024 pairs by ONE WAVE (four buckets per workgroup, no workgroup barrier), up to 4096 by
//                            the workgroup; larger buckets were put on a list by the scan kernel ...
//   seg_l2_prep_kernel + the pass kernels again   ... and become the segments of a second-level LSD sort over their
//                            remaining bits (persistent grids: a request without such buckets pays a few empty launches)
// MODE 0 / 3 run ceil(rbits / 8) global LSD passes -- ceil(rbits / 9) with 9-bit digits where that is one pass fewer --
//        (ascending (table, row, position) order, like round 2's segmented sort);
// MODE 1 partitions on the LOW row digit (order (table, row & 255, row >> 8, position): buckets balanced under any skew, but
//        neighbours in the sorted array are not neighbours in the table -- the apply kernel pays for that under skew);
// MODE 2 partitions on the TOP row digit (ascending order; a skewed head makes its bucket a second-level segment).
// Equal keys end up adjacent and in request order in every mode -- all the apply kernel needs.
#include <cstdlib>

#include "common.h"
#include "pm_experiments.h"

namespace pm {
namespace {

constexpr int kT = 256;                 // threads per workgroup
constexpr int kWaves = kT / kWave;      // 4
constexpr int kTile = kSegTile;         // elements per radix tile (16 per thread); common.h
constexpr int kTileItems = kTile / kT;  // 16
constexpr int kRadix = 256;
constexpr int kRadixMax = 512;          // mode 0 sorts 9 bits per pass where that saves a pass
constexpr uint32_t kWaveCap = 1024;     // bucket-local sort by one wave: 16 pairs per lane
constexpr uint32_t kLocalCap = kTile;   // ... by one workgroup; larger buckets go to the second level
constexpr int kL2Grid = 512;            // persistent grid of the second-level passes
constexpr int kBuildBags = 1024;        // bags per workgroup of the key-building kernel
constexpr int kMaxLbPasses = 5;         // passes the look-back form of mode 0 takes at most

struct SegHeader {
    uint32_t n_total;    // pairs in all segments (= length of the sorted arrays)
    uint32_t n_tiles;    // radix tiles of the first level
    uint32_t n_l2;       // buckets on the second-level list (= second-level segments)
    uint32_t n_tiles2;   // radix tiles of the second level
    uint32_t lookback_timeouts;   // look-back walks that stopped waiting and counted a predecessor tile's digits themselves (a statistic)
    uint32_t rest_pairs;          // hybrid backward: flagged lookups / tables hyb_rest_kernel finished in LDS (zeroed by prep 1; statistics)
    uint32_t rest_tables;
    uint32_t pad[9];
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

// digit of pass `pass` for a segment that sorts `bits` bits starting at bit `bit0`: (shift, mask).  mode 0 / 1: LSD, rb (8 or 9)
// bits per pass from bit0 (mode 1 runs pass 0 only).  mode 2: pass 0 = the top digit.  A width of 0 makes the pass a stable copy.
// rb = 9 (512 digit values, mode 0 only) is chosen where it saves a global pass: 25 ... 27 row bits (the 40 M-row Criteo tables:
// 3 passes instead of 4), 17 / 18 bits (2 instead of 3), 9 bits.
__device__ __forceinline__ void pass_digit(int mode, int pass, uint32_t rbits_packed, int rb, int& shift, uint32_t& mask) {
    const int bits = static_cast<int>(rbits_packed & 255u), bit0 = static_cast<int>(rbits_packed >> 8);
    int w;
    if (mode == 2) {
        w = bits < 8 ? bits : 8;
        shift = bit0 + bits - w;
    } else {
        shift = bit0 + rb * pass;
        w = bits - rb * pass;
        w = w < 0 ? 0 : (w > rb ? rb : w);
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

// (match_digit: common.h)

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

// (wave_lds_fence: common.h)

// ---------------------------------------------------------------------------------------------------------------------
// prep 1: one workgroup (1024 threads) per table
__global__ void __launch_bounds__(1024) seg_prep_tables_kernel(const void* indices, const void* offsets, int idx64, const int64_t* rows, int T,
                                                               int64_t B, int64_t N, int64_t bag_begin, int64_t bag_count, int force_ragged,
                                                               SegDesc* desc, uint32_t* zero4, const HybArgs hyb, HybTable* hyb_tab, uint32_t* qtail,
                                                               uint32_t* rest_stat) {
    __shared__ uint32_t s_sample[2048];      // classification: 2^16 hashed bits
    __shared__ uint32_t s_rep;
    const int t = blockIdx.x;
    if (zero4 && t == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0u;
    if (t == 0 && threadIdx.x < 2) rest_stat[threadIdx.x] = 0u;
    if (qtail && t < kHybMaxTables && static_cast<int>(threadIdx.x) < hyb.slices) qtail[static_cast<size_t>(t) * hyb.slices + threadIdx.x] = 0u;   // hyb_part_kernel's queue lengths
    const int64_t TB = static_cast<int64_t>(T) * B;
    const int64_t g0 = static_cast<int64_t>(t) * B + bag_begin;
    auto off_at = [&](int64_t g) -> int64_t { return g < TB ? load_index(offsets, g, idx64) : N; };
    const int64_t s = off_at(g0);
    const int64_t e = off_at(g0 + bag_count);
    const int64_t cnt = e > s ? e - s : 0;
    const int64_t L = (bag_count > 0 && cnt > 0 && cnt % bag_count == 0) ? cnt / bag_count : 0;
    int bad = (L == 0 || force_ragged) ? 1 : 0;
    // Classification for the hybrid backward (common.h), first half: is the table structurally eligible -- a size the dup map
    // separates, a row count that makes repeats rare under uniform indices -- and if so, request a sample of 2048 lookups spread
    // over the table's slice now, so that their latency passes under the check of the offsets below.
    const int64_t n_rows = rows[t];
    // (lookups <= rows / 4: under uniform indices at most ~22 % of the lookups then sit in rows looked up twice -- the N = 8 rank
    //  shape, 1.3 M lookups into 10 M rows, has 12 %; round 4's rows / 8 excluded it)
    bool cand = t < kHybMaxTables && !bad && cnt >= kHybMinCount && cnt <= static_cast<int64_t>(hyb.slices) * (kHybMaxCount / kBloomK) &&
                cnt * 4 <= n_rows;     // workgroup-uniform
    const bool sample = cand && hyb.allow == 1;                // allow == 2 (tests): structural eligibility is enough; 0: nothing to decide
    uint32_t sr[2] = {0u, 0u};
    if (sample) {
        const int64_t stride = cnt / 2048;
#pragma unroll
        for (int u = 0; u < 2; ++u) sr[u] = static_cast<uint32_t>(load_index(indices, s + (threadIdx.x + u * 1024) * stride, idx64));
        for (int i = threadIdx.x; i < 2048; i += 1024) s_sample[i] = 0u;
        if (threadIdx.x == 0) s_rep = 0u;
    }
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
    // ... second half: the sample must not repeat itself (2^16 hashed bits: uniform indices collide ~32 times in 2048, a
    // Zipf(1.05) head hundreds of times).  A wrong verdict costs speed only: every lookup is applied exactly once either way.
    {
        cand = cand && !bad;
        if (sample) {
            uint32_t rep = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t h = (sr[u] * 0x9E3779B1u) >> 16;
                rep += (atomicOr(&s_sample[h >> 5], 1u << (h & 31u)) >> (h & 31u)) & 1u;
            }
            if (rep) atomicAdd(&s_rep, rep);
            __syncthreads();
            cand = cand && s_rep * 8u <= 2048u;
        }
        if (threadIdx.x == 0) {
            HybTable h;
            h.mode = 0u;                                      // (final verdict: hyb_mark_kernel)
            h.cand = (cand && hyb.allow) ? 1u : 0u;
            h.pooling = bad ? 0u : static_cast<uint32_t>(L);
            h.n_dup = 0u;
            hyb_tab[t] = h;
        }
    }
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
    if (!WEIGHTED && (desc[t].pooling > 0 || desc[t].pad)) return;      // pad: a hybrid table, its pairs are compacted already
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
        hdr->lookback_timeouts = 0;
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

// one digit histogram update of a wave (the per-pass histogram kernel of mode 3): a wave whose keys all share the digit adds
// once (top digits of a skewed head, small tables); else one LDS atomic per lane.  The all-pass histogram of mode 0 does better
// (seg_hist_all_kernel: hot digits found once per wave, counted by ballot) -- same-address LDS atomics are served one lane after
// the other, and that WAS what bounded it under a skewed request.
__device__ __forceinline__ void wave_hist_add(uint32_t* h, uint32_t dg, bool valid, int lane) {
    const uint64_t vmask = __ballot(valid);
    const uint32_t firstd = __builtin_amdgcn_readfirstlane(dg);
    const bool uniform = __ballot(valid && dg != firstd) == 0 && (vmask & 1ull);
    if (uniform) {
        if (lane == 0) atomicAdd(&h[firstd], static_cast<uint32_t>(__popcll(vmask)));
    } else if (valid) {
        atomicAdd(&h[dg], 1u);
    }
}

// the row ids (or keys: their low bits are the row) of a tile for the histogram kernels: position k * 256 + thread, all loads
// in flight before the first use
template <typename K>
__device__ __forceinline__ void load_tile_rows(const TileDesc& td, const PassSrc<K>& src, uint64_t (&row)[kTileItems]) {
    const uint32_t cnt = td.cnt, last = cnt - 1u;       // cnt >= 1; positions past the end re-read the last element (and are ignored)
    auto at = [&](int k) { const uint32_t q = k * kT + threadIdx.x; return q < cnt ? q : last; };
    if (src.first && td.pooling > 0) {
        // request-order tiles of a table with a pooling factor: the rows are the index array's (tables whose pairs were built --
        // ragged bags, weighted requests, hybrid tables' compacted repeats -- read the built keys below: same row bits, and a
        // hybrid table's pairs are not at their request positions any more)
        if (src.idx64) {
            const PM_GLOBAL int64_t* ip = as_global<int64_t>(src.indices) + td.in_base;
#pragma unroll
            for (int k = 0; k < kTileItems; ++k) row[k] = static_cast<uint64_t>(ip[at(k)]);
        } else {
            const PM_GLOBAL int32_t* ip = as_global<int32_t>(src.indices) + td.in_base;
            uint32_t raw[kTileItems];
#pragma unroll
            for (int k = 0; k < kTileItems; ++k) raw[k] = static_cast<uint32_t>(ip[at(k)]);
#pragma unroll
            for (int k = 0; k < kTileItems; ++k) row[k] = raw[k];
        }
    } else {
        const K* kp = src.keys + (src.first ? td.in_base : td.out_base);
        K raw[kTileItems];
#pragma unroll
        for (int k = 0; k < kTileItems; ++k) raw[k] = kp[at(k)];
#pragma unroll
        for (int k = 0; k < kTileItems; ++k) row[k] = static_cast<uint64_t>(raw[k]);
    }
}

// tiles are taken g = blockIdx.x, + gridDim.x, ... < *n_tiles: level 1 launches one workgroup per possible tile, level 2 a
// persistent grid (the tile count of level 2 is known only on the device, and is zero for most requests)
template <typename K, int RB>
__global__ void __launch_bounds__(kT) seg_hist_kernel(const TileDesc* tiles, const uint32_t* n_tiles, const PassSrc<K> src, int mode, int pass,
                                                      uint32_t* bh) {
    constexpr int RAD = 1 << RB;
    __shared__ uint32_t h[RAD];
    // the first descriptor is fetched together with the tile count, not after it (grids never exceed the descriptor arrays):
    // a workgroup's start-up is a chain of dependent loads, and every link costs a memory latency that nothing hides
    TileDesc td = tiles[blockIdx.x];
    const uint32_t nt = *n_tiles;
    for (uint32_t g = blockIdx.x; g < nt; g += gridDim.x) {
        if (g != blockIdx.x) td = tiles[g];
        const uint32_t cnt = td.cnt;
        int shift;
        uint32_t mask;
        pass_digit(mode, pass, td.rbits, RB, shift, mask);
        for (int i = threadIdx.x; i < RAD; i += kT) h[i] = 0;
        __syncthreads();
        const int lane = threadIdx.x % kWave;
        uint64_t row[kTileItems];
        load_tile_rows<K>(td, src, row);
#pragma unroll
        for (int k = 0; k < kTileItems; ++k) {
            const uint32_t i = static_cast<uint32_t>(k) * kT + threadIdx.x;
            const bool valid = i < cnt;
            const uint32_t dg = valid ? static_cast<uint32_t>(row[k] >> shift) & mask : 0u;
            wave_hist_add(h, dg, valid, lane);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < RAD; i += kT) bh[static_cast<uint64_t>(g) * RAD + i] = h[i];
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
template <int RB>
__global__ void __launch_bounds__(kRadix * kScanChunks) seg_scan_kernel(SegHeader* hdr, const SegDesc* desc, const uint32_t* n_seg_dev,
                                                                        uint32_t n_seg_host, uint32_t* bh, uint32_t* bstart, uint32_t* bcnt,
                                                                        int classify, int mode, uint32_t* l2_list) {
    constexpr int RAD = 1 << RB;
    constexpr int DPT = RAD / kRadix;                    // digits per thread: 1 (256 digit values) or 2 (512: d and d + 256)
    __shared__ uint32_t s_sum[kScanChunks][RAD];
    __shared__ uint32_t s_tmp[kWaves];
    const int d0 = threadIdx.x % kRadix;
    const int c = threadIdx.x / kRadix;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    const uint32_t n_seg = n_seg_dev ? *n_seg_dev : n_seg_host;
    for (uint32_t t = blockIdx.x; t < n_seg; t += gridDim.x) {
        const SegDesc sd = desc[t];
        const uint64_t r0 = sd.tile_base;
        const uint32_t per = (sd.ntiles + kScanChunks - 1) / kScanChunks;
        const uint32_t ra = c * per < sd.ntiles ? c * per : sd.ntiles;
        const uint32_t rb = ra + per < sd.ntiles ? ra + per : sd.ntiles;
        // a thread's digits (d0 and, with 512 digit values, d0 + 256) walk the tiles together: the loads of both are in
        // flight at once, the chain of dependent trips stays as short as with one digit
        {
            uint32_t sum[DPT];
#pragma unroll
            for (int j = 0; j < DPT; ++j) sum[j] = 0;
            for (uint32_t r = ra; r < rb; r += kScanU) {
                uint32_t v[DPT][kScanU];
#pragma unroll
                for (int u = 0; u < kScanU; ++u)
#pragma unroll
                    for (int j = 0; j < DPT; ++j) v[j][u] = (r + u < rb) ? bh[(r0 + r + u) * RAD + d0 + j * kRadix] : 0u;
#pragma unroll
                for (int u = 0; u < kScanU; ++u)
#pragma unroll
                    for (int j = 0; j < DPT; ++j) sum[j] += v[j][u];
            }
#pragma unroll
            for (int j = 0; j < DPT; ++j) s_sum[c][d0 + j * kRadix] = sum[j];
        }
        __syncthreads();
        uint32_t total[DPT];
        {
            uint32_t run[DPT];
#pragma unroll
            for (int j = 0; j < DPT; ++j) {
                run[j] = 0;
                total[j] = 0;
#pragma unroll
                for (int cc = 0; cc < kScanChunks; ++cc) {
                    const uint32_t x = s_sum[cc][d0 + j * kRadix];
                    if (cc < c) run[j] += x;
                    total[j] += x;
                }
            }
            for (uint32_t r = ra; r < rb; r += kScanU) {     // in-place rewrite: the loads of a batch are issued before its stores
                uint32_t v[DPT][kScanU];
#pragma unroll
                for (int u = 0; u < kScanU; ++u)
#pragma unroll
                    for (int j = 0; j < DPT; ++j) v[j][u] = (r + u < rb) ? bh[(r0 + r + u) * RAD + d0 + j * kRadix] : 0u;
#pragma unroll
                for (int u = 0; u < kScanU; ++u)
#pragma unroll
                    for (int j = 0; j < DPT; ++j) {
                        if (r + u < rb) bh[(r0 + r + u) * RAD + d0 + j * kRadix] = run[j];
                        run[j] += v[j][u];
                    }
            }
        }
        // exclusive scan of the digit totals in digit order (the 256 threads of chunk 0; everybody keeps the barriers): the
        // digits d0 of all threads first, then the digits d0 + 256
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            uint32_t incl = (c == 0) ? total[j] : 0u;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, kWave);
                if (lane >= off) incl += up;
            }
            if (c == 0 && lane == kWave - 1) s_tmp[wave] = incl;
            __syncthreads();
            uint32_t all = 0;
            for (int w = 0; w < kWaves; ++w) all += s_tmp[w];
            if (c == 0) {
                uint32_t base = 0;
                for (int w = 0; w < wave; ++w) base += s_tmp[w];
                const uint32_t b = t * RAD + d0 + j * kRadix;
                bstart[b] = sd.out_start + carry + base + incl - total[j];
                bcnt[b] = total[j];
                if (classify && total[j] > kLocalCap) {
                    int lo, hi;
                    local_bits(mode, static_cast<int>(sd.rbits), lo, hi);
                    if (hi > lo) l2_list[atomicAdd(&hdr->n_l2, 1u)] = b;
                }
            }
            carry += all;
            __syncthreads();     // s_tmp (and, after the last round, s_sum) are rewritten next
        }
    }
}

// Stable placement of a tile's elements by one digit, staged in LDS.  The elements sit in registers: thread (wave, lane)
// holds tile positions wave * chunk + r * 64 + lane, r = 0 .. ITEMS-1 (valid: r * 64 + lane < chunk and position < cnt),
// so waves own consecutive runs of the tile and (wave, r, lane) order = position order.  On return s_key / s_val hold
// the tile reordered by digit (stable), s_dstart[d] = first staged position of digit d.
template <typename K, int ITEMS, int RB>
__device__ __forceinline__ void tile_count_digits(const K (&key)[ITEMS], uint32_t (&rank)[ITEMS], uint32_t cnt, uint32_t chunk, int shift,
                                                  uint32_t mask, uint32_t* s_wcnt) {
    constexpr int RAD = 1 << RB;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    for (int i = threadIdx.x; i < kWaves * RAD; i += kT) s_wcnt[i] = 0;
    __syncthreads();
    uint32_t* wcnt = s_wcnt + wave * RAD;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        rank[r] = 0;
        if (static_cast<uint32_t>(r) * kWave < chunk) {     // wave-uniform
            const uint32_t off = static_cast<uint32_t>(r) * kWave + lane;
            const bool valid = off < chunk && wave * chunk + off < cnt;
            const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & mask;
            uint32_t below, total;
            match_digit<RB>(d, valid, below, total);
            const uint32_t base = valid ? wcnt[d] : 0u;
            rank[r] = base + below;
            // the lowest lane of each match set advances the digit's counter; a wave executes its LDS operations in program
            // order, so the next round's reads see it
            if (valid && below == 0) wcnt[d] = base + total;
        }
    }
    __syncthreads();
}

// ... per-wave counts -> per-wave starts inside each digit's staged run, s_dstart[d] = first staged position of digit d; with
// s_count, the tile's count of every digit is left there.  Ends with a barrier.
template <int RB>
__device__ __forceinline__ void tile_digit_starts(uint32_t* s_wcnt, uint32_t* s_dstart, uint32_t* s_tmp, uint32_t* s_count) {
    constexpr int RAD = 1 << RB;
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < RAD / kT; ++j) {              // digit order: the digits 0 .. 255 of all threads, then 256 .. 511
        const int d = threadIdx.x + j * kT;
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t c = s_wcnt[w * RAD + d];
            s_wcnt[w * RAD + d] = acc;     // wave w's elements of digit d start this far into the digit's staged run
            acc += c;
        }
        if (s_count) s_count[d] = acc;
        s_dstart[d] = carry + block_excl_scan256(acc, s_tmp);
        if (j + 1 < RAD / kT) {
            carry += s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
            __syncthreads();               // s_tmp is rewritten by the next scan
        }
    }
    __syncthreads();
}

// ... the elements go to their staged positions.  Ends with a barrier.
template <typename K, int ITEMS, int RB>
__device__ __forceinline__ void tile_place(const K (&key)[ITEMS], const uint32_t (&val)[ITEMS], const uint32_t (&rank)[ITEMS], uint32_t cnt,
                                           uint32_t chunk, int shift, uint32_t mask, K* s_key, uint32_t* s_val, const uint32_t* s_wcnt,
                                           const uint32_t* s_dstart, bool barrier) {
    constexpr int RAD = 1 << RB;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    const uint32_t* wcnt = s_wcnt + wave * RAD;
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
    if (barrier) __syncthreads();
}

template <typename K, int ITEMS, int RB = 8>
__device__ __forceinline__ void tile_stage_by_digit(const K (&key)[ITEMS], const uint32_t (&val)[ITEMS], uint32_t cnt, uint32_t chunk,
                                                    int shift, uint32_t mask, K* s_key, uint32_t* s_val, uint32_t* s_wcnt,
                                                    uint32_t* s_dstart, uint32_t* s_tmp) {
    uint32_t rank[ITEMS];
    tile_count_digits<K, ITEMS, RB>(key, rank, cnt, chunk, shift, mask, s_wcnt);
    tile_digit_starts<RB>(s_wcnt, s_dstart, s_tmp, nullptr);
    tile_place<K, ITEMS, RB>(key, val, rank, cnt, chunk, shift, mask, s_key, s_val, s_wcnt, s_dstart, true);
}

// a tile's pairs into registers: thread (wave, lane) holds tile positions wave * 1024 + r * 64 + lane.  ALL loads are issued
// before the first value is used (written as one loop, load and key arithmetic per element, the compiler waits for every
// load before it issues the next: sixteen memory latencies in a row at the start of every workgroup).
template <typename K>
__device__ __forceinline__ void load_tile_pairs(const TileDesc& td, const PassSrc<K>& src, uint64_t base, bool from_idx, K (&key)[kTileItems],
                                                uint32_t (&val)[kTileItems]) {
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    constexpr uint32_t chunk = kTile / kWaves;
    const uint32_t cnt = td.cnt, p0 = wave * chunk + lane, last = cnt - 1u;     // cnt >= 1; positions past the end re-read the last
    auto at = [&](int r) { const uint32_t q = p0 + r * kWave; return q < cnt ? q : last; };   // element: straight-line code
    if (from_idx) {
        if (src.idx64) {
            const PM_GLOBAL int64_t* ip = as_global<int64_t>(src.indices) + base;
            int64_t raw[kTileItems];
#pragma unroll
            for (int r = 0; r < kTileItems; ++r) raw[r] = ip[at(r)];
#pragma unroll
            for (int r = 0; r < kTileItems; ++r) key[r] = static_cast<K>(raw[r]);
        } else {
            const PM_GLOBAL int32_t* ip = as_global<int32_t>(src.indices) + base;
            int32_t raw[kTileItems];
#pragma unroll
            for (int r = 0; r < kTileItems; ++r) raw[r] = ip[at(r)];
#pragma unroll
            for (int r = 0; r < kTileItems; ++r) key[r] = static_cast<K>(static_cast<uint32_t>(raw[r]));
        }
        const K tkey = static_cast<K>(td.seg) << src.tshift;
#pragma unroll
        for (int r = 0; r < kTileItems; ++r) {
            const uint32_t pos = p0 + r * kWave;
            const bool valid = pos < cnt;
            key[r] = valid ? (tkey | key[r]) : static_cast<K>(0);
            val[r] = valid ? src.bag_begin + fast_div(td.first + pos, td.pooling, td.magic) : 0u;
        }
    } else {
#pragma unroll
        for (int r = 0; r < kTileItems; ++r) {
            const uint32_t pos = p0 + r * kWave;
            const bool valid = pos < cnt;
            const K kk = src.keys[base + at(r)];
            const uint32_t vv = src.vals[base + at(r)];
            key[r] = valid ? kk : static_cast<K>(0);
            val[r] = valid ? vv : 0u;
        }
    }
}

// one tile of a scatter pass (inlined into both kernels below: the LDS arrays keep their address space)
template <typename K, int RB>
__device__ __forceinline__ void scatter_tile(const TileDesc td, uint32_t g, const PassSrc<K>& src, int mode, int pass, const uint32_t* prefix,
                                             const uint32_t* bstart, K* kout, uint32_t* vout, K* s_key, uint32_t* s_val, uint32_t* s_wcnt,
                                             uint32_t* s_dstart, uint32_t* s_gbase, uint32_t* s_tmp, uint32_t lane_zero) {
    constexpr uint32_t chunk = kTile / kWaves;
    const uint32_t t = td.seg, cnt = td.cnt;
    int shift;
    uint32_t mask;
    constexpr int RAD = 1 << RB;
    pass_digit(mode, pass, td.rbits, RB, shift, mask);
    const uint64_t base = (src.first ? td.in_base : td.out_base) + lane_zero;
    const bool from_idx = src.first && td.pooling > 0;    // keys formed from the index array, bag = position / pooling
#pragma unroll
    for (int d = threadIdx.x; d < RAD; d += kT) s_gbase[d] = bstart[t * RAD + d] + prefix[static_cast<uint64_t>(g) * RAD + d];
    K key[kTileItems];
    uint32_t val[kTileItems];
    load_tile_pairs<K>(td, src, base, from_idx, key, val);
    tile_stage_by_digit<K, kTileItems, RB>(key, val, cnt, chunk, shift, mask, s_key, s_val, s_wcnt, s_dstart, s_tmp);
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
template <typename K, int RB>
__global__ void __launch_bounds__(kT) seg_scatter_kernel(const TileDesc* tiles, const uint32_t* n_tiles, const PassSrc<K> src, int mode, int pass,
                                                         const uint32_t* prefix, const uint32_t* bstart, K* kout, uint32_t* vout) {
    __shared__ K s_key[kTile];
    __shared__ uint32_t s_val[kTile];
    __shared__ uint32_t s_wcnt[kWaves << RB];
    __shared__ uint32_t s_dstart[1 << RB];
    __shared__ uint32_t s_gbase[1 << RB];
    __shared__ uint32_t s_tmp[kWaves];
    const TileDesc td = tiles[blockIdx.x];      // together with the tile count (see seg_hist_kernel)
    if (blockIdx.x >= *n_tiles) return;
    scatter_tile<K, RB>(td, blockIdx.x, src, mode, pass, prefix, bstart, kout, vout, s_key, s_val, s_wcnt, s_dstart, s_gbase, s_tmp, 0u);
}

// level 2: persistent grid.  (A plain loop around the tile body lets the compiler hoist every lane-dependent address out of
// it -- 179 VGPRs, two workgroups per CU --; the body's addresses therefore hang on a zero the compiler cannot see through.)
template <typename K, int RB>
__global__ void __launch_bounds__(kT) seg_scatter_loop_kernel(const TileDesc* tiles, const uint32_t* n_tiles, const PassSrc<K> src, int mode,
                                                              int pass, const uint32_t* prefix, const uint32_t* bstart, K* kout, uint32_t* vout) {
    __shared__ K s_key[kTile];
    __shared__ uint32_t s_val[kTile];
    __shared__ uint32_t s_wcnt[kWaves << RB];
    __shared__ uint32_t s_dstart[1 << RB];
    __shared__ uint32_t s_gbase[1 << RB];
    __shared__ uint32_t s_tmp[kWaves];
    const uint32_t nt = *n_tiles;
#pragma clang loop unroll(disable)
    for (uint32_t g = blockIdx.x; g < nt; g += gridDim.x) {
        uint32_t lane_zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
        scatter_tile<K, RB>(tiles[g], g, src, mode, pass, prefix, bstart, kout, vout, s_key, s_val, s_wcnt, s_dstart, s_gbase, s_tmp, lane_zero);
        __syncthreads();   // s_gbase / the staged tile are rewritten by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Mode 0 in ONE kernel per pass ("look-back" form; the three-kernel pass above stays for modes 1 / 2, the second level and
// requests of 2^30 pairs or more).  The digit histograms of ALL passes are taken in one read of the request (seg_hist_all),
// one small kernel turns them into the tables' bucket starts for every pass (seg_scan_all: table totals do not depend on
// the order of the pairs), and each pass is then a single launch in which a tile learns how many pairs of each digit
// precede it in its table from the tiles before it WHILE THE PASS RUNS: every tile publishes its digit counts in a status
// row (one 32-bit word per digit: 2 flag bits | 30 value bits), walks back over its predecessors' rows adding their
// counts until it meets one that already carries an inclusive prefix, and publishes its own inclusive prefix.  Rows are
// written and polled with device-coherent (sc1) 16-byte accesses, four digits per lane; a word is self-contained (value
// and flag travel together), so no fence is involved.  A tile only ever waits for tiles of lower index in the same table.
// PROGRESS (round 4): a walk that has polled an unpublished predecessor kLbSpinCap times stops waiting and counts that tile's
// digits itself -- the tile's pairs are the previous pass's output, complete since the kernel boundary -- so every workgroup can
// finish on its own whatever the order in which the others are dispatched or run (HIP promises none); the result is the same
// either way, hdr->lookback_timeouts merely counts such walks.  Round 4 first took the tile ids from atomic tickets instead
// (rocPRIM's ordered block id): one counter per pass queued ~2000 atomics on a word (+9 us per pass), per-table counters put a
// returning atomic in front of every tile's loads (+7 us per pass), and requesting the statically assigned tile's pairs under the
// atomic failed because ~1000 workgroups start at once and draw their tickets in any order (97 % reloads, +16 us per pass).
// Counting for a predecessor costs nothing until it happens.
constexpr uint32_t kStAggregate = 1u << 30, kStInclusive = 2u << 30, kStValue = (1u << 30) - 1u;
constexpr size_t kLbRowWords = 4 * 512;   // passes x digit values of any plan the look-back form takes (seg_sort_lookback): 4 x 9 bits, 5 x 8 bits
constexpr uint32_t kLbSpinCap = 1u << 12;    // polls (>= ~1 us each) before a walk counts for its predecessor

struct U4 { uint32_t x, y, z, w; };
using u32x4 = __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t;

__device__ __forceinline__ u32x4 load_status(const uint32_t* p) {      // device-coherent: never served from a stale line
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
#ifndef PM_LB_BATCH
#define PM_LB_BATCH 4
#endif
constexpr int kLbBatch = PM_LB_BATCH;       // predecessors' rows requested per trip of the walk (8: measured no faster, round 4)
__device__ __forceinline__ void load_status_batch(const uint32_t* const (&p)[kLbBatch], u32x4 (&v)[kLbBatch]) {
#if PM_LB_BATCH == 8
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "memory");
#else
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
        : "memory");
#endif
}
// `s_nop 1`: the data registers of a store wider than 64 bits must not be written by the next VALU instruction (the wait state the
// compiler pads for stores it can see; it cannot see this one)
__device__ __forceinline__ void store_status(uint32_t* p, u32x4 v) {   // write-through
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// all passes' digit counts of every tile of the REQUEST: status rows st[pass][tile][digit] = count, flagged as published for
// pass 0 only (pass 0's tiles are the request's; the other rows give seg_scan_all the table totals and read "not published")
template <typename K, int RB>
__global__ void __launch_bounds__(kT) seg_hist_all_kernel(const TileDesc* tiles, const uint32_t* n_tiles, const PassSrc<K> src, int npass,
                                                          uint32_t tiles_cap, uint32_t* st) {
    constexpr int RAD = 1 << RB;
    __shared__ uint32_t h[kMaxLbPasses * RAD];
    const TileDesc td = tiles[blockIdx.x];
    const uint32_t nt = *n_tiles;
    if (blockIdx.x >= nt) return;
    const uint32_t g = blockIdx.x, cnt = td.cnt;
    int shift[kMaxLbPasses];
    uint32_t mask[kMaxLbPasses];
#pragma unroll
    for (int p = 0; p < kMaxLbPasses; ++p) pass_digit(0, p, td.rbits, RB, shift[p], mask[p]);
    for (int i = threadIdx.x; i < npass * RAD; i += kT) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x % kWave;
    uint64_t row[kTileItems];
    load_tile_rows<K>(td, src, row);
    // Hot digits.  Lanes of a wave that add to the same counter are served one after the other, and under a skewed request most
    // lanes do: the top digit of a Zipf(1.05) head is one value for ~70 % of the lookups, and the histogram then ran 40 us against
    // 14 with the atomics alone removed from the uniform request's 20 (tools/r4_sort_probe.py, round 4).  So every wave looks at
    // its first element of each pass -- up to three digits, those of the first lanes not yet accounted for -- and remembers up to
    // two digits that eight lanes or more share; for the whole tile, lanes holding such a digit are counted with one ballot into a
    // scalar register and added once, by one lane, at the end.  A wave that finds none (uniform indices: always) pays nothing per element: plain LDS atomics, which
    // is also what the all-lanes-equal test of wave_hist_add cost the uniform request 6 us for.
    constexpr uint32_t kNone = 0xffffffffu;
    uint32_t hot0[kMaxLbPasses], hot1[kMaxLbPasses];
#pragma unroll
    for (int p = 0; p < kMaxLbPasses; ++p) {
        hot0[p] = hot1[p] = kNone;
        if (p < npass) {
            const bool v0 = threadIdx.x < cnt;
            const uint32_t dg = static_cast<uint32_t>(row[0] >> shift[p]) & mask[p];
            uint64_t cand = __ballot(v0);
            for (int r = 0; r < 3 && cand != 0; ++r) {
                const int leader = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(cand)) - 1);
                const uint32_t d0 = __builtin_amdgcn_readlane(dg, leader);
                const uint64_t same = __ballot(v0 && dg == d0);
                cand &= ~same;
                if (__popcll(same) >= 8) {
                    if (hot0[p] == kNone) hot0[p] = d0;
                    else if (hot1[p] == kNone) hot1[p] = d0;
                }
            }
        }
    }
    bool any_hot = false;
#pragma unroll
    for (int p = 0; p < kMaxLbPasses; ++p) any_hot = any_hot || hot0[p] != kNone;
    if (!any_hot) {                                                       // wave-uniform
#pragma unroll
        for (int k = 0; k < kTileItems; ++k) {
            const bool valid = static_cast<uint32_t>(k) * kT + threadIdx.x < cnt;
#pragma unroll
            for (int p = 0; p < kMaxLbPasses; ++p)
                if (p < npass && valid) atomicAdd(&h[p * RAD + (static_cast<uint32_t>(row[k] >> shift[p]) & mask[p])], 1u);
        }
    } else {
        uint32_t c0[kMaxLbPasses], c1[kMaxLbPasses];                      // the wave's lanes that held a hot digit: scalar counts, added once
#pragma unroll
        for (int p = 0; p < kMaxLbPasses; ++p) c0[p] = c1[p] = 0u;
#pragma unroll
        for (int k = 0; k < kTileItems; ++k) {
            const bool valid = static_cast<uint32_t>(k) * kT + threadIdx.x < cnt;
#pragma unroll
            for (int p = 0; p < kMaxLbPasses; ++p) {
                if (p < npass) {
                    const uint32_t dg = static_cast<uint32_t>(row[k] >> shift[p]) & mask[p];
                    const bool e0 = dg == hot0[p], e1 = dg == hot1[p];    // kNone equals no digit
                    c0[p] += static_cast<uint32_t>(__popcll(__ballot(valid && e0)));
                    c1[p] += static_cast<uint32_t>(__popcll(__ballot(valid && e1)));
                    if (valid && !e0 && !e1) atomicAdd(&h[p * RAD + dg], 1u);
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < kMaxLbPasses; ++p) {
                if (p < npass) {
                    if (c0[p]) atomicAdd(&h[p * RAD + (hot0[p] & (RAD - 1))], c0[p]);
                    if (c1[p]) atomicAdd(&h[p * RAD + (hot1[p] & (RAD - 1))], c1[p]);
                }
            }
        }
    }
    __syncthreads();
    const bool first_tile = td.first == 0;
    for (int i = threadIdx.x; i < npass * RAD; i += kT) {
        const int p = i / RAD, d = i % RAD;
        const uint32_t c = h[i];
        // pass 0: the count is this tile's aggregate (for a table's first tile: its inclusive prefix) from the start.  Later
        // passes: flag 0 = not published -- the count is there for seg_scan_all only (these tiles are not those passes' tiles)
        st[(static_cast<uint64_t>(p) * tiles_cap + g) * RAD + d] = c | (p != 0 ? 0u : first_tile ? kStInclusive : kStAggregate);
    }
}

// grid (T, npass): bucket starts of table t in pass p = the table's first output position + exclusive scan over the digits of
// the column sums of its tiles' counts
template <int RB>
__global__ void __launch_bounds__(kRadix * kScanChunks) seg_scan_all_kernel(const SegDesc* desc, const uint32_t* st, uint32_t tiles_cap, int T,
                                                                            uint32_t* bstart_all) {
    constexpr int RAD = 1 << RB;
    constexpr int DPT = RAD / kRadix;
    __shared__ uint32_t s_sum[kScanChunks][RAD];
    __shared__ uint32_t s_tmp[kWaves];
    const int d0 = threadIdx.x % kRadix, c = threadIdx.x / kRadix;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    const uint32_t t = blockIdx.x, p = blockIdx.y;
    const SegDesc sd = desc[t];
    const uint64_t r0 = static_cast<uint64_t>(p) * tiles_cap + sd.tile_base;
    const uint32_t per = (sd.ntiles + kScanChunks - 1) / kScanChunks;
    const uint32_t ra = c * per < sd.ntiles ? c * per : sd.ntiles;
    const uint32_t rb = ra + per < sd.ntiles ? ra + per : sd.ntiles;
    uint32_t sum[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) sum[j] = 0;
    for (uint32_t r = ra; r < rb; r += kScanU) {
        // rows past the chunk re-read its last row and are dropped afterwards: straight-line loads, all in flight together (a
        // load under a predicate is waited for where it stands)
        uint32_t v[DPT][kScanU];
#pragma unroll
        for (int u = 0; u < kScanU; ++u) {
            const uint32_t rr = r + u < rb ? r + u : rb - 1u;
#pragma unroll
            for (int j = 0; j < DPT; ++j) v[j][u] = st[(r0 + rr) * RAD + d0 + j * kRadix];
        }
#pragma unroll
        for (int u = 0; u < kScanU; ++u)
#pragma unroll
            for (int j = 0; j < DPT; ++j) sum[j] += (r + u < rb) ? (v[j][u] & kStValue) : 0u;
    }
#pragma unroll
    for (int j = 0; j < DPT; ++j) s_sum[c][d0 + j * kRadix] = sum[j];
    __syncthreads();
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        uint32_t total = 0;
#pragma unroll
        for (int cc = 0; cc < kScanChunks; ++cc) total += s_sum[cc][d0 + j * kRadix];
        uint32_t incl = (c == 0) ? total : 0u;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += up;
        }
        if (c == 0 && lane == kWave - 1) s_tmp[wave] = incl;
        __syncthreads();
        uint32_t all = 0;
        for (int w = 0; w < kWaves; ++w) all += s_tmp[w];
        if (c == 0) {
            uint32_t base = 0;
            for (int w = 0; w < wave; ++w) base += s_tmp[w];
            bstart_all[(static_cast<uint64_t>(p) * T + t) * RAD + d0 + j * kRadix] = sd.out_start + carry + base + incl - total;
        }
        carry += all;
        __syncthreads();
    }
}

// one pass: grid = one workgroup per possible tile
template <typename K, int RB>
__global__ void __launch_bounds__(kT) seg_lookback_pass_kernel(const TileDesc* tiles, SegHeader* hdr, const PassSrc<K> src, int pass, int T,
                                                               uint32_t tiles_cap, uint32_t* st_all, const uint32_t* bstart_all, K* kout,
                                                               uint32_t* vout, uint32_t spin_cap) {
    constexpr int RAD = 1 << RB;
    constexpr int LW = RAD / 256;            // waves that publish and walk back: 256 digits each, four per lane
    __shared__ K s_key[kTile];
    __shared__ uint32_t s_val[kTile];
    __shared__ uint32_t s_wcnt[kWaves << RB];
    __shared__ uint32_t s_dstart[1 << RB];
    __shared__ __attribute__((aligned(16))) uint32_t s_gbase[1 << RB];
    __shared__ uint32_t s_tmp[kWaves];
    __shared__ uint32_t s_fb[LW * 256];      // fallback of the walk: a predecessor's digit counts, recomputed
    const TileDesc td = tiles[blockIdx.x];
    if (blockIdx.x >= hdr->n_tiles) return;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    constexpr uint32_t chunk = kTile / kWaves;
    const uint32_t g = blockIdx.x, t = td.seg, cnt = td.cnt, first = td.first;
    const uint32_t j = first / static_cast<uint32_t>(kTile);                  // this tile's index inside its table
    int shift;
    uint32_t mask;
    pass_digit(0, pass, td.rbits, RB, shift, mask);
    uint32_t* st = st_all + static_cast<uint64_t>(pass) * tiles_cap * RAD;
    const int dq = (wave * 256 + lane * 4) & (RAD - 1);                       // the look-back waves' first digit
    u32x4 tb = {0u, 0u, 0u, 0u};
    if (wave < LW) tb = *reinterpret_cast<const u32x4*>(bstart_all + (static_cast<uint64_t>(pass) * T + t) * RAD + dq);
    const uint64_t base = src.first ? td.in_base : td.out_base;
    const bool from_idx = src.first && td.pooling > 0;    // keys formed from the index array, bag = position / pooling
    K key[kTileItems];
    uint32_t val[kTileItems];
    const unsigned trace_id = static_cast<unsigned>(pass) * 4096u + blockIdx.x;      // (experiment builds: PM_STAMP; nothing otherwise)
    (void)trace_id;
    PM_STAMP(trace_id, 0);
    load_tile_pairs<K>(td, src, base, from_idx, key, val);
    PM_STAMP_DRAINED(trace_id, 1);
    uint32_t rank[kTileItems];
    tile_count_digits<K, kTileItems, RB>(key, rank, cnt, chunk, shift, mask, s_wcnt);
    PM_STAMP(trace_id, 2);
    tile_digit_starts<RB>(s_wcnt, s_dstart, s_tmp, s_gbase);                  // s_gbase: the tile's count of every digit, for now
    u32x4 mine = {0u, 0u, 0u, 0u};
    if (wave < LW) {
        mine = *reinterpret_cast<const u32x4*>(s_gbase + dq);
        // pass 0's aggregates were published by seg_hist_all; a table's first tile publishes its inclusive prefix at once
        if (pass != 0) store_status(st + static_cast<uint64_t>(g) * RAD + dq, mine | (j == 0 ? kStInclusive : kStAggregate));
    }
    PM_STAMP_DRAINED(trace_id, 3);
    tile_place<K, kTileItems, RB>(key, val, rank, cnt, chunk, shift, mask, s_key, s_val, s_wcnt, s_dstart, false);
    PM_STAMP(trace_id, 4);
    if (wave < LW) {
        u32x4 ex = {0u, 0u, 0u, 0u};
        u32x4 open = {~0u, ~0u, ~0u, ~0u};                                    // digits still walking: all-ones
        bool walking = j > 0;
        uint32_t k = j;                                                       // next predecessor: table tile k - 1
        while (walking) {
            // kLbBatch predecessors per trip, their rows requested together (short of predecessors, the table's first tile is
            // read again and skipped); the walk waits only at a row that is not published yet
            const uint32_t* row[kLbBatch];
            u32x4 s4[kLbBatch];
#pragma unroll
            for (int b = 0; b < kLbBatch; ++b) {
                const uint32_t kb = k > static_cast<uint32_t>(b) ? k - 1u - b : 0u;
                row[b] = st + static_cast<uint64_t>(g - (j - kb)) * RAD + dq;
            }
            load_status_batch(row, s4);
#pragma unroll
            for (int b = 0; b < kLbBatch; ++b) {
                if (walking && k > 0) {                                       // wave-uniform
                    --k;
                    u32x4 v = s4[b];
                    uint32_t spins = 0;
                    const bool force = spin_cap == 0xffffffffu && pass != 0;      // tests: every predecessor is counted here, none is believed
                    while (force || __any(((v[0] >> 30) == 0u) | ((v[1] >> 30) == 0u) | ((v[2] >> 30) == 0u) | ((v[3] >> 30) == 0u))) {
                        if (force || ++spins > spin_cap) {
                            // The predecessor has not published: stop waiting and COUNT ITS DIGITS HERE.  Its pairs are the previous
                            // pass's output (complete: kernel boundary), so this wave can always finish its walk on its own -- the
                            // sort's progress rests on no assumption about which workgroups run when, and a walk never gives up.
                            // (pass 0 never gets here: its counts were published by the histogram kernel before this launch.)
                            const TileDesc pd = tiles[g - (j - k)];
                            uint32_t* fb = s_fb + (wave % LW) * 256;
#pragma unroll
                            for (int i = 0; i < 4; ++i) fb[lane * 4 + i] = 0u;
                            wave_lds_fence();
                            for (uint32_t q = lane; q < pd.cnt; q += kWave) {
                                const uint32_t dg = static_cast<uint32_t>(src.keys[static_cast<uint64_t>(pd.out_base) + q] >> shift) & mask;
                                if ((dg >> 8) == static_cast<uint32_t>(wave % LW) || RAD == 256) atomicAdd(&fb[dg & 255u], 1u);
                            }
                            wave_lds_fence();
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = fb[lane * 4 + c] | kStAggregate;
                            if (lane == 0) atomicAdd(&hdr->lookback_timeouts, 1u);     // a statistic: walks that counted for a predecessor
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        v = load_status(row[b]);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        ex[c] += v[c] & kStValue & open[c];
                        if ((v[c] >> 30) == 2u) open[c] = 0u;
                    }
                    walking = k > 0 && __any((open[0] | open[1] | open[2] | open[3]) != 0u);
                }
            }
        }
        if (j > 0) store_status(st + static_cast<uint64_t>(g) * RAD + dq, ((ex + mine) & kStValue) | kStInclusive);
        *reinterpret_cast<u32x4*>(s_gbase + dq) = tb + ex;                    // where this tile's run of each digit starts
        PM_STAMP_DRAINED(trace_id, 5);
    }
    __syncthreads();
    PM_STAMP(trace_id, 6);
#pragma unroll 4
    for (int q0 = 0; q0 < kTileItems; ++q0) {
        const uint32_t q = q0 * kT + threadIdx.x;
        if (q < cnt) {
            const K kk = s_key[q];
            const uint32_t dg = static_cast<uint32_t>(kk >> shift) & mask;
            const uint64_t o = static_cast<uint64_t>(s_gbase[dg]) + (q - s_dstart[dg]);
            kout[o] = kk;
            vout[o] = s_val[q];
        }
    }
    PM_STAMP_DRAINED(trace_id, 7);
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
                uint32_t below, total;
                match_digit<8>(d, valid, below, total);
                const uint32_t base = valid ? wcnt[d] : 0u;
                rank[r] = base + below;
                if (valid && below == 0) wcnt[d] = base + total;
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

#include "seg_hybrid.inc"

inline size_t a256(size_t x) { return (x + 255) / 256 * 256; }
inline size_t tiles_max(size_t n, int T) { return n / kTile + static_cast<size_t>(T) + 1; }
inline size_t seg2_max(size_t n) { return n / kTile + 1; }
// per-tile counts of the bag-major apply: a hybrid table has at most kHybMaxCount lookups, a tile at least 4 bags of >= 1 lookup;
// tables of more tiles than this are not compacted (the launcher falls back to "no hybrid tables")
inline size_t tile_cnt_stride(size_t) { return kCompactMaxTiles; }                 // every second-level segment holds more than a tile
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
    uint32_t* st_all;        // look-back form: status rows [pass][tile][digit]
    uint32_t* bstart_all;    // ... bucket starts [pass][table][digit]
    HybTable* hyb_tab;       // hybrid backward: per-table records
    uint32_t* bloom;         // ... dup bitmaps of the first kHybMaxTables tables
    uint32_t* tile_cnt;      // ... flagged lookups per tile of the bag-major apply, [T_h][tile_cnt_stride]
    uint32_t* qtail;         // ... rows dealt to the queue of (table, slice), [T_h][slices]
    uint32_t* rest_stage;    // ... staged left-overs of the tables hyb_rest_kernel finishes, [T_h][2][kRestCap]
    uint32_t* rest_n;        // ... their numbers, [T_h]
    size_t total;
};

// hyb = the request can take the hybrid backward (unweighted): only then does the layout END with the dup maps (256 KB .. 4 MB per
// table: up to 512 MB at 128 tables), the per-tile counts, the slice queues' lengths and the left-over stage.  Everything before them
// sits where it sits either way, so the accessors below need not know (weighted requests used to reserve all of it).
Scratch scratch_layout(void* base, size_t n, int T, bool hyb = true) {
    Scratch s;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = p ? p + off : nullptr; off += a256(bytes); return q; };
    const size_t tm = tiles_max(n, T), nb = static_cast<size_t>(T) * kRadix, s2 = seg2_max(n), t2 = tiles2_max(n);
    s.hdr = reinterpret_cast<SegHeader*>(take(sizeof(SegHeader)));
    s.desc = reinterpret_cast<SegDesc*>(take(sizeof(SegDesc) * static_cast<size_t>(T)));
    s.tiles = reinterpret_cast<TileDesc*>(take(sizeof(TileDesc) * tm));
    s.bh = reinterpret_cast<uint32_t*>(take(4 * tm * kRadixMax));       // level 1 may run 9-bit digits (seg_sort_radix_bits)
    s.bstart = reinterpret_cast<uint32_t*>(take(4 * static_cast<size_t>(T) * kRadixMax));
    s.bcnt = reinterpret_cast<uint32_t*>(take(4 * static_cast<size_t>(T) * kRadixMax));
    s.l2_list = reinterpret_cast<uint32_t*>(take(4 * nb));
    s.desc2 = reinterpret_cast<SegDesc*>(take(sizeof(SegDesc) * s2));
    s.tiles2 = reinterpret_cast<TileDesc*>(take(sizeof(TileDesc) * t2));
    s.bh2 = reinterpret_cast<uint32_t*>(take(4 * t2 * kRadix));
    s.bstart2 = reinterpret_cast<uint32_t*>(take(4 * s2 * kRadix));
    s.bcnt2 = reinterpret_cast<uint32_t*>(take(4 * s2 * kRadix));
    s.st_all = reinterpret_cast<uint32_t*>(take(4 * tm * kLbRowWords));
    s.bstart_all = reinterpret_cast<uint32_t*>(take(4 * static_cast<size_t>(T) * kLbRowWords));
    const size_t th = static_cast<size_t>(T < kHybMaxTables ? T : kHybMaxTables);
    s.hyb_tab = reinterpret_cast<HybTable*>(take(sizeof(HybTable) * static_cast<size_t>(T)));
    const size_t hy = hyb ? 1 : 0;
    s.bloom = reinterpret_cast<uint32_t*>(take(hy * 4 * th * static_cast<size_t>(hyb_slices(static_cast<int64_t>(n), T)) * kBloomWords));
    s.tile_cnt = reinterpret_cast<uint32_t*>(take(hy * 4 * th * tile_cnt_stride(n)));
    s.qtail = reinterpret_cast<uint32_t*>(take(hy * 4 * th * static_cast<size_t>(kBloomKMax)));
    s.rest_stage = reinterpret_cast<uint32_t*>(take(hy * 4 * th * 2 * static_cast<size_t>(kRestCap)));
    s.rest_n = reinterpret_cast<uint32_t*>(take(hy * 4 * th));
    s.total = off;
    return s;
}

}  // namespace


size_t seg_sort_scratch_bytes(size_t n_max, int T, bool hybrid_possible) { return scratch_layout(nullptr, n_max, T, hybrid_possible).total; }

const SegDesc* seg_sort_desc(const void* scratch, size_t n_max, int T) { return scratch_layout(const_cast<void*>(scratch), n_max, T).desc; }
const uint32_t* seg_sort_count(const void* scratch, size_t n_max, int T) {
    return &scratch_layout(const_cast<void*>(scratch), n_max, T).hdr->n_total;
}
const uint32_t* seg_sort_timeouts(const void* scratch, size_t n_max, int T) {
    return &scratch_layout(const_cast<void*>(scratch), n_max, T).hdr->lookback_timeouts;
}
const HybTable* seg_sort_hyb_tab(const void* scratch, size_t n_max, int T) { return scratch_layout(const_cast<void*>(scratch), n_max, T).hyb_tab; }
const uint32_t* seg_sort_bloom(const void* scratch, size_t n_max, int T) { return scratch_layout(const_cast<void*>(scratch), n_max, T).bloom; }
uint32_t* seg_sort_tile_cnt(const void* scratch, size_t n_max, int T) { return scratch_layout(const_cast<void*>(scratch), n_max, T).tile_cnt; }
size_t seg_sort_tile_cnt_stride(size_t n_max) { return tile_cnt_stride(n_max); }
uint32_t* seg_sort_rest_stat(void* scratch, size_t n_max, int T) { return &scratch_layout(scratch, n_max, T).hdr->rest_pairs; }
const uint32_t* seg_sort_rest_stage(const void* scratch, size_t n_max, int T) { return scratch_layout(const_cast<void*>(scratch), n_max, T).rest_stage; }
const uint32_t* seg_sort_rest_n(const void* scratch, size_t n_max, int T) { return scratch_layout(const_cast<void*>(scratch), n_max, T).rest_n; }

// mode 0 sorts 9 bits per pass where that saves a global pass over the pairs: 25 .. 27 row bits (the 40 M-row Criteo tables:
// 3 passes instead of 4), 17 / 18 bits (2 instead of 3), 9 bits.  Everywhere else 8: a 512-value digit costs LDS (three
// workgroups per CU instead of four) and twice the per-tile histogram traffic.
int seg_sort_radix_bits(int mode, int rbits_max) {
    if ((mode != 0 && mode != 3) || rbits_max <= 8) return 8;
    return (rbits_max + 8) / 9 < (rbits_max + 7) / 8 ? 9 : 8;
}
int seg_sort_passes(int mode, int rbits_max) {
    if ((mode != 0 && mode != 3) || rbits_max <= 0) return 1;
    const int rb = seg_sort_radix_bits(mode, rbits_max);
    return (rbits_max + rb - 1) / rb;
}
bool seg_sort_result_in_b(int mode, int rbits_max) { return seg_sort_passes(mode, rbits_max) % 2 == 1; }

// the look-back form serves mode 0 where its status words (30 value bits) and scratch rows (kLbRowWords) hold the request
bool seg_sort_lookback(int mode, int rbits_max, int64_t n) {
    if (mode != 0 || n >= (1ll << 30)) return false;        // mode 3 = mode 0 with the three-kernel passes
    const int np = seg_sort_passes(mode, rbits_max);
    return np <= kMaxLbPasses && static_cast<size_t>(np) << seg_sort_radix_bits(mode, rbits_max) <= kLbRowWords;
}

namespace {
template <typename K, int RB>
void launch_lookback(const Scratch& s, unsigned tm, int T, PassSrc<K> src, int total, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                     uint32_t spin_cap, hipStream_t stream) {
    src.first = 1;
    src.keys = keys_a;
    src.vals = vals_a;
    hipLaunchKernelGGL((seg_hist_all_kernel<K, RB>), dim3(tm), dim3(kT), 0, stream, s.tiles, &s.hdr->n_tiles, src, total, tm, s.st_all);
    hipLaunchKernelGGL((seg_scan_all_kernel<RB>), dim3(T, total), dim3(kRadix * kScanChunks), 0, stream, s.desc, s.st_all, tm, T, s.bstart_all);
    for (int p = 0; p < total; ++p) {
        src.first = p == 0 ? 1 : 0;
        src.keys = p == 0 ? keys_a : (p % 2 == 1 ? keys_b : keys_a);
        src.vals = p == 0 ? vals_a : (p % 2 == 1 ? vals_b : vals_a);
        K* kout = (p % 2 == 0) ? keys_b : keys_a;
        uint32_t* vout = (p % 2 == 0) ? vals_b : vals_a;
        hipLaunchKernelGGL((seg_lookback_pass_kernel<K, RB>), dim3(tm), dim3(kT), 0, stream, s.tiles, s.hdr, src, p, T, tm, s.st_all, s.bstart_all,
                           kout, vout, spin_cap);
    }
}
template <typename K, int RB>
void launch_level1_pass(const Scratch& s, unsigned tm, int T, const PassSrc<K>& src, int mode, int p, K* kout, uint32_t* vout, hipStream_t stream) {
    hipLaunchKernelGGL((seg_hist_kernel<K, RB>), dim3(tm), dim3(kT), 0, stream, s.tiles, &s.hdr->n_tiles, src, mode, p, s.bh);
    hipLaunchKernelGGL((seg_scan_kernel<RB>), dim3(T), dim3(kRadix * kScanChunks), 0, stream, s.hdr, s.desc, static_cast<const uint32_t*>(nullptr),
                       static_cast<uint32_t>(T), s.bh, s.bstart, s.bcnt, (mode != 0 && p == 0) ? 1 : 0, mode, s.l2_list);
    hipLaunchKernelGGL((seg_scatter_kernel<K, RB>), dim3(tm), dim3(kT), 0, stream, s.tiles, &s.hdr->n_tiles, src, mode, p, s.bh, s.bstart, kout, vout);
}
}  // namespace

// The mark kernel needs 128 KB of dynamic LDS: asked for once per process; where it cannot be had (another ARCH, a smaller
// carve-out) the hybrid path is simply not offered -- the sort then runs complete, as for any other request.
bool seg_sort_hybrid_available() {
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(hyb_mark_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               2 * kBloomWords * 4) == hipSuccess;
    return ok;
}

// part A: the tables' segments, pooling factors and verdicts; the dup bitmaps of the hybrid tables
template <typename K>
hipError_t seg_sort_part_a(const SegSortRequest& rq, void* scratch, hipStream_t stream) {
    if (rq.N == 0) return hipSuccess;
    if (rq.T < 1 || rq.T > kSegSortMaxTables || rq.N > 0xffffffffLL) return hipErrorInvalidValue;
    const size_t n = static_cast<size_t>(rq.N);
    const Scratch s = scratch_layout(scratch, n, rq.T);
    HybArgs hyb = rq.hyb;
    if (rq.weighted) hyb.allow = 0;
    hipLaunchKernelGGL(seg_prep_tables_kernel, dim3(rq.T), dim3(1024), 0, stream, rq.indices, rq.offsets, rq.idx64, rq.rows, rq.T, rq.B, rq.N,
                       rq.bag_begin, rq.bag_count, rq.weighted ? 1 : 0, s.desc, rq.zero4, hyb, s.hyb_tab, hyb.allow ? s.qtail : nullptr,
                       &s.hdr->rest_pairs);
    if (hyb.allow) {
        if (!seg_sort_hybrid_available()) return hipErrorInvalidValue;      // (sort_indices asks first and does not offer the path then)
        const int th = rq.T < kHybMaxTables ? rq.T : kHybMaxTables;
        // deal the candidate tables' rows out to per-slice queues first (hyb_part_kernel): PARAM_AMD_HYB_PART = 0 never, 1 always,
        // else (default) from `kPartMinSlices` slices per table on -- see the measurements in seg_hybrid.inc
        static const int part_env = [] { const char* e = getenv("PARAM_AMD_HYB_PART"); return e ? atoi(e) : -1; }();          // once per process
        const bool part = rq.queue_a && rq.queue_b && part_env != 0 && (part_env == 1 || hyb.slices >= kPartMinSlices);
        if (part) {
            const unsigned pgrid = static_cast<unsigned>((rq.N + kPartChunk - 1) / kPartChunk);
#define PM_LAUNCH_PART(NB_)                                                                                                               \
    hipLaunchKernelGGL((hyb_part_kernel<NB_>), dim3(pgrid), dim3(kMarkThreads), 0, stream, s.desc, s.hyb_tab, rq.indices, rq.idx64, th, rq.N, \
                       hyb.slices, rq.queue_a, rq.queue_b, s.qtail)
            switch (hyb.slices) {
                case 4: PM_LAUNCH_PART(2); break;
                case 8: PM_LAUNCH_PART(3); break;
                case 16: PM_LAUNCH_PART(4); break;
                case 32: PM_LAUNCH_PART(5); break;
                default: PM_LAUNCH_PART(6); break;
            }
#undef PM_LAUNCH_PART
        }
        const unsigned grid = static_cast<unsigned>(kXcds * hyb.slices * ((th + kXcds - 1) / kXcds));
        hipLaunchKernelGGL(hyb_mark_kernel, dim3(grid), dim3(kMarkThreads), 2 * kBloomWords * 4, stream, s.desc, s.hyb_tab, rq.indices, rq.idx64, th,
                           rq.T, rq.N, s.bloom, hyb.slices, part ? rq.queue_a : nullptr, part ? rq.queue_b : nullptr, part ? s.qtail : nullptr);
    }
    return hipGetLastError();
}

// between the bag-major apply and part B of a hybrid sort: stage or compact what the apply listed (hyb_stage_kernel)
template <typename K>
hipError_t seg_sort_stage_leftovers(const SegSortRequest& rq, const K* keys_b, const uint32_t* vals_b, K* keys_a, uint32_t* vals_a,
                                    HybTiles tiles, int rest_enable, void* scratch, hipStream_t stream) {
    if (rq.N == 0) return hipSuccess;
    if (rq.T < 1 || rq.T > kSegSortMaxTables || rq.N > 0xffffffffLL) return hipErrorInvalidValue;
    if (tiles.tiles_per_table < 1 || tiles.tiles_per_table > kCompactMaxTiles) return hipErrorInvalidValue;
    const size_t n = static_cast<size_t>(rq.N);
    const Scratch s = scratch_layout(scratch, n, rq.T);
    const int th = rq.T < kHybMaxTables ? rq.T : kHybMaxTables;
    const int parts = stage_parts(tiles.tiles_per_table);
    hipLaunchKernelGGL((hyb_stage_kernel<K>), dim3(th * parts), dim3(kT), 0, stream, s.desc, s.hyb_tab, th, s.tile_cnt, tile_cnt_stride(n), tiles,
                       keys_b, vals_b, keys_a, vals_a, parts, rest_enable, rq.tshift, s.rest_stage, s.rest_n, &s.hdr->rest_pairs);
    return hipGetLastError();
}

template <typename K>
hipError_t seg_sort_part_b(const SegSortRequest& rq, int mode, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* bag_of,
                           void* scratch, hipStream_t stream, bool hybrid_done) {
    if (rq.N == 0) return hipSuccess;
    if (rq.T < 1 || rq.T > kSegSortMaxTables || rq.N > 0xffffffffLL) return hipErrorInvalidValue;
    const size_t n = static_cast<size_t>(rq.N);
    const Scratch s = scratch_layout(scratch, n, rq.T);
    const unsigned tm = static_cast<unsigned>(tiles_max(n, rq.T));
    // a hybrid sort's part B follows the bag-major kernel and hyb_stage_kernel (seg_sort_stage_leftovers), which has turned every
    // hybrid table's segment into "no pairs" (staged for hyb_rest_kernel's LDS sort) or "n built pairs" (compacted into the a buffers)
    if (rq.hyb.allow && !rq.weighted && !hybrid_done) return hipErrorInvalidValue;
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
    const int total = seg_sort_passes(mode, rq.rbits_max), rb = seg_sort_radix_bits(mode, rq.rbits_max);
    const bool lookback = seg_sort_lookback(mode, rq.rbits_max, rq.N);
    if (mode == 3) mode = 0;
    PassSrc<K> src;
    src.indices = rq.indices;
    src.idx64 = rq.idx64;
    src.tshift = rq.tshift;
    src.bag_begin = static_cast<uint32_t>(rq.bag_begin);
    if (lookback) {
        const uint32_t cap = rq.spin_cap ? rq.spin_cap : kLbSpinCap;
        if (rb == 9) launch_lookback<K, 9>(s, tm, rq.T, src, total, keys_a, keys_b, vals_a, vals_b, cap, stream);
        else launch_lookback<K, 8>(s, tm, rq.T, src, total, keys_a, keys_b, vals_a, vals_b, cap, stream);
        return hipGetLastError();
    }
    for (int p = 0; p < total; ++p) {
        src.first = p == 0 ? 1 : 0;
        src.keys = p == 0 ? keys_a : (p % 2 == 1 ? keys_b : keys_a);
        src.vals = p == 0 ? vals_a : (p % 2 == 1 ? vals_b : vals_a);
        K* kout = (p % 2 == 0) ? keys_b : keys_a;
        uint32_t* vout = (p % 2 == 0) ? vals_b : vals_a;
        if (rb == 9) launch_level1_pass<K, 9>(s, tm, rq.T, src, mode, p, kout, vout, stream);
        else launch_level1_pass<K, 8>(s, tm, rq.T, src, mode, p, kout, vout, stream);
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
            hipLaunchKernelGGL((seg_hist_kernel<K, 8>), dim3(kL2Grid), dim3(kT), 0, stream, s.tiles2, &s.hdr->n_tiles2, src, 0, q, s.bh2);
            hipLaunchKernelGGL((seg_scan_kernel<8>), dim3(kL2Grid / 2), dim3(kRadix * kScanChunks), 0, stream, s.hdr, s.desc2, &s.hdr->n_l2, 0u, s.bh2, s.bstart2,
                               s.bcnt2, 0, 0, static_cast<uint32_t*>(nullptr));
            hipLaunchKernelGGL((seg_scatter_loop_kernel<K, 8>), dim3(kL2Grid), dim3(kT), 0, stream, s.tiles2, &s.hdr->n_tiles2, src, 0, q, s.bh2,
                               s.bstart2, kout, vout);
        }
    }
    return hipGetLastError();
}

template <typename K>
hipError_t seg_sort_pairs(const SegSortRequest& rq, int mode, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* bag_of,
                          void* scratch, hipStream_t stream) {
    if (rq.hyb.allow) return hipErrorInvalidValue;      // the hybrid form's part B belongs to the apply call
    const hipError_t rc = seg_sort_part_a<K>(rq, scratch, stream);
    if (rc != hipSuccess) return rc;
    return seg_sort_part_b<K>(rq, mode, keys_a, keys_b, vals_a, vals_b, bag_of, scratch, stream, false);
}

#define PM_SEG_INST(K_)                                                                                                              \
    template hipError_t seg_sort_part_a<K_>(const SegSortRequest&, void*, hipStream_t);                                              \
    template hipError_t seg_sort_part_b<K_>(const SegSortRequest&, int, K_*, K_*, uint32_t*, uint32_t*, uint32_t*, void*, hipStream_t, bool); \
    template hipError_t seg_sort_stage_leftovers<K_>(const SegSortRequest&, const K_*, const uint32_t*, K_*, uint32_t*, HybTiles, int, void*,    \
                                                     hipStream_t);                                                                              \
    template hipError_t seg_sort_pairs<K_>(const SegSortRequest&, int, K_*, K_*, uint32_t*, uint32_t*, uint32_t*, void*, hipStream_t);
PM_SEG_INST(uint32_t)
PM_SEG_INST(uint64_t)
#undef PM_SEG_INST

}  // namespace pm

PM_DEFINE_TRACE_READER(pm_experiment_trace)      // experiment builds only (pm_experiments.h); nothing in the product
