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
// Kernels (256 threads, gfx950, no inter-workgroup communication inside a kernel -- kernel boundaries are the only
// synchronisation, as in radix_sort.hip):
//   seg_prep_tables_kernel   one workgroup per table: segment start / count from the offsets, pooling factor if every bag
//                            of the (sliced) table has the same length, key bits of the table from rows[t]
//   seg_prep_scan_kernel     one workgroup: output start and first tile of every segment, tile -> segment map, header
//   seg_build_keys_kernel    only for tables WITHOUT a pooling factor (ragged, weighted): (key, bag) per lookup at request
//                            positions (binary search over LDS-staged offsets); workgroups of other tables exit at once
//   seg_hist / seg_scan / seg_scatter   one radix pass over every segment's 4096-element tiles: per-tile digit counts,
//                            per-segment exclusive prefix (+ absolute bucket starts), stable scatter
//   seg_local_sort_kernel    MODE 1 / 2: after ONE partition pass every (table, digit) bucket is sorted by its remaining
//                            key bits inside LDS (<= 2048 elements: 8 per thread in registers; <= 10240: 40 per thread,
//                            one workgroup per CU; larger: a single-workgroup external sort through the spare buffers)
// MODE 0 runs ceil(rbits / 8) global LSD passes (ascending (table, row, position) order, like round 2's segmented sort);
// MODE 1 partitions on the LOW row digit (balanced buckets under any skew; order (table, row & 255, row >> 8, position));
// MODE 2 partitions on the TOP row digit (ascending order; a skewed head makes its bucket large).
// Equal keys end up adjacent and in request order in every mode -- all the apply kernel needs.
#include <cstdlib>

#include "common.h"

namespace pm {
namespace {

constexpr int kT = 256;                 // threads per workgroup
constexpr int kWaves = kT / kWave;      // 4
constexpr int kTile = 4096;             // elements per radix tile (16 per thread)
constexpr int kTileItems = kTile / kT;
constexpr int kRadix = 256;
constexpr int kSmallItems = 8;          // bucket-local sort, small class: <= 2048 elements, registers
constexpr int kBigItems = 40;           // big class: <= 10240 elements, one workgroup per CU (85 KB of LDS)
constexpr uint32_t kSmallCap = kSmallItems * kT;
constexpr uint32_t kBigCap = kBigItems * kT;

struct SegHeader {
    uint32_t n_total;   // elements in all segments (= length of the sorted arrays)
    uint32_t n_tiles;   // radix tiles in all segments
    uint32_t n_big;     // buckets in the big list
    uint32_t n_huge;    // buckets in the huge list
    uint32_t pad[12];
};

__device__ __forceinline__ int bits_for_dev(uint64_t n_values) {   // bits needed to represent 0 .. n_values - 1
    return n_values <= 1 ? 0 : 64 - __builtin_clzll(n_values - 1);
}

// digit of pass `pass` for a table whose rows need `rbits` bits: (shift, width).  mode 0: LSD, 8 bits per pass from bit 0.
// mode 1: pass 0 = the low digit.  mode 2: pass 0 = the top digit.  A width of 0 makes the pass a stable copy.
__device__ __forceinline__ void pass_digit(int mode, int pass, int rbits, int& shift, uint32_t& mask) {
    int w;
    if (mode == 2) {
        w = rbits < 8 ? rbits : 8;
        shift = rbits - w;
    } else {
        shift = 8 * pass;
        w = rbits - shift;
        w = w < 0 ? 0 : (w > 8 ? 8 : w);
        if (w == 0) shift = 0;
    }
    mask = (1u << w) - 1u;
}
// bits the bucket-local sort still has to order: [lo, hi)
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

// ---------------------------------------------------------------------------------------------------------------------
// prep 1: one workgroup per table
__global__ void __launch_bounds__(kT) seg_prep_tables_kernel(const void* offsets, int idx64, const int64_t* rows, int T, int64_t B,
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
        for (int64_t i = threadIdx.x; i < bag_count; i += kT)
            if (off_at(g0 + i) != s + i * L) bad = 1;
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

// prep 2: one workgroup of 1024 threads (T <= 1024): exclusive scans over the tables, tile -> segment map, header
__global__ void __launch_bounds__(1024) seg_prep_scan_kernel(SegDesc* desc, int T, SegHeader* hdr, uint32_t* tile_seg) {
    __shared__ uint32_t s_cnt[1024], s_til[1024], s_tb[1025];
    const int t = threadIdx.x;
    const uint32_t c = t < T ? desc[t].count : 0u;
    const uint32_t nt = t < T ? desc[t].ntiles : 0u;
    s_cnt[t] = c;
    s_til[t] = nt;
    __syncthreads();
    // Hillis-Steele inclusive scans in LDS (one launch per sort, 1024 values: not worth anything cleverer)
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t a = t >= off ? s_cnt[t - off] : 0u;
        const uint32_t b = t >= off ? s_til[t - off] : 0u;
        __syncthreads();
        s_cnt[t] += a;
        s_til[t] += b;
        __syncthreads();
    }
    const uint32_t out_start = s_cnt[t] - c, tile_base = s_til[t] - nt;
    const uint32_t n_total = s_cnt[1023], n_tiles = s_til[1023];
    __syncthreads();
    if (t < T) {
        desc[t].out_start = out_start;
        desc[t].tile_base = tile_base;
        s_tb[t] = tile_base;
    }
    if (t == 0) {
        s_tb[T] = n_tiles;
        hdr->n_total = n_total;
        hdr->n_tiles = n_tiles;
        hdr->n_big = 0;
        hdr->n_huge = 0;
    }
    __syncthreads();
    for (uint32_t g = t; g < n_tiles; g += 1024) {
        int lo = 0, hi = T;             // largest t with s_tb[t] <= g (segments without tiles repeat their neighbour's base)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_tb[mid] <= g) lo = mid; else hi = mid;
        }
        tile_seg[g] = static_cast<uint32_t>(lo);
    }
}

// keys / values at request positions for the tables that have no pooling factor (and, WEIGHTED, for all: the value is the
// lookup's position, its bag goes to bag_of).  grid (bag tiles of 256, T).
template <typename K, bool WEIGHTED>
__global__ void __launch_bounds__(kT) seg_build_keys_kernel(const void* indices, const void* offsets, int idx64, int T, int64_t B,
                                                            int64_t N, int64_t bag_begin, int64_t bag_count, const SegDesc* desc,
                                                            int tshift, K* keys, uint32_t* vals, uint32_t* bag_of) {
    constexpr int kBags = 256;
    __shared__ int64_t s_off[kBags + 1];
    const int t = blockIdx.y;
    if (!WEIGHTED && desc[t].pooling > 0) return;
    const int64_t bag0 = bag_begin + static_cast<int64_t>(blockIdx.x) * kBags;
    const int64_t left = bag_begin + bag_count - bag0;
    if (left <= 0) return;
    const int nb = left < kBags ? static_cast<int>(left) : kBags;
    const int64_t TB = static_cast<int64_t>(T) * B;
    const int64_t g0 = static_cast<int64_t>(t) * B + bag0;
    for (int i = threadIdx.x; i <= nb; i += kT) s_off[i] = (g0 + i < TB) ? load_index(offsets, g0 + i, idx64) : N;
    __syncthreads();
    const int64_t base = s_off[0], end = s_off[nb];
    for (int64_t j = base + threadIdx.x; j < end; j += kT) {
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

// ---------------------------------------------------------------------------------------------------------------------
// What a pass reads: pass 0 the request (index array, or the built keys of tables without a pooling factor) at request
// positions; later passes the previous pass's output at output positions.
template <typename K>
struct PassSrc {
    const void* indices;    // pass 0
    int idx64;
    const K* keys;          // pass 0: built keys (request positions); later: previous output
    const uint32_t* vals;
    int first;              // 1: pass 0
    int tshift;
    uint32_t bag_begin;
};

template <typename K>
__global__ void __launch_bounds__(kT) seg_hist_kernel(const SegHeader* hdr, const SegDesc* desc, const uint32_t* tile_seg,
                                                      const PassSrc<K> src, int mode, int pass, uint32_t* bh) {
    __shared__ uint32_t h[kRadix];
    const uint32_t g = blockIdx.x;
    if (g >= hdr->n_tiles) return;
    const uint32_t t = tile_seg[g];
    const SegDesc d = desc[t];
    const uint32_t first = (g - d.tile_base) * static_cast<uint32_t>(kTile);
    const uint32_t cnt = (d.count - first) < static_cast<uint32_t>(kTile) ? d.count - first : static_cast<uint32_t>(kTile);
    int shift;
    uint32_t mask;
    pass_digit(mode, pass, static_cast<int>(d.rbits), shift, mask);
    h[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x % kWave;
    const uint64_t base = static_cast<uint64_t>(src.first ? d.in_start : d.out_start) + first;
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
}

// per segment: exclusive prefix of the tile counts per digit (in place), absolute start and size of every (segment, digit)
// bucket; classify != 0 (bucket-local sort follows): buckets too large for the small class go to the big / huge lists.
__global__ void __launch_bounds__(kRadix) seg_scan_kernel(SegHeader* hdr, const SegDesc* desc, uint32_t* bh, uint32_t* bstart,
                                                          uint32_t* bcnt, int classify, int mode, uint32_t* big_list,
                                                          uint32_t* huge_list) {
    __shared__ uint32_t s_tmp[kWaves];
    const int d = threadIdx.x;
    const uint32_t t = blockIdx.x;
    const SegDesc sd = desc[t];
    const uint64_t r0 = sd.tile_base;
    uint32_t run = 0;
    for (uint32_t r = 0; r < sd.ntiles; r += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (r + u < sd.ntiles) ? bh[(r0 + r + u) * kRadix + d] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (r + u < sd.ntiles) bh[(r0 + r + u) * kRadix + d] = run;
            run += v[u];
        }
    }
    const uint32_t start = sd.out_start + block_excl_scan256(run, s_tmp);
    const uint32_t b = t * kRadix + d;
    bstart[b] = start;
    bcnt[b] = run;
    if (classify && run > kSmallCap) {
        int lo, hi;
        local_bits(mode, static_cast<int>(sd.rbits), lo, hi);
        if (hi > lo) {
            if (run <= kBigCap) big_list[atomicAdd(&hdr->n_big, 1u)] = b;
            else huge_list[atomicAdd(&hdr->n_huge, 1u)] = b;
        }
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

template <typename K>
__global__ void __launch_bounds__(kT) seg_scatter_kernel(const SegHeader* hdr, const SegDesc* desc, const uint32_t* tile_seg,
                                                         const PassSrc<K> src, int mode, int pass, const uint32_t* prefix,
                                                         const uint32_t* bstart, K* kout, uint32_t* vout) {
    __shared__ K s_key[kTile];
    __shared__ uint32_t s_val[kTile];
    __shared__ uint32_t s_wcnt[kWaves * kRadix];
    __shared__ uint32_t s_dstart[kRadix];
    __shared__ uint32_t s_gbase[kRadix];
    __shared__ uint32_t s_tmp[kWaves];
    const uint32_t g = blockIdx.x;
    if (g >= hdr->n_tiles) return;
    const uint32_t t = tile_seg[g];
    const SegDesc d = desc[t];
    const uint32_t first = (g - d.tile_base) * static_cast<uint32_t>(kTile);
    const uint32_t cnt = (d.count - first) < static_cast<uint32_t>(kTile) ? d.count - first : static_cast<uint32_t>(kTile);
    int shift;
    uint32_t mask;
    pass_digit(mode, pass, static_cast<int>(d.rbits), shift, mask);
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    constexpr uint32_t chunk = kTile / kWaves;
    const uint64_t base = static_cast<uint64_t>(src.first ? d.in_start : d.out_start) + first;
    const bool from_idx = src.first && d.pooling > 0;     // keys formed from the index array, bag = position / pooling
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
                val[r] = src.bag_begin + (first + pos) / d.pooling;
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

// ---------------------------------------------------------------------------------------------------------------------
// bucket-local sort (in place in the b buffers): the bucket's pairs are loaded into registers, ordered by their remaining
// key bits with stable 8-bit rounds through LDS, and written back as one contiguous run.
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
    __syncthreads();             // the LDS tile is reused by the next bucket of a looping workgroup
}

// a bucket too large for LDS: one workgroup sorts it by its remaining bits with stable 8-bit passes streamed through
// global memory, ping-pong between the bucket's range of the b buffers and the same range of the (by now unused) a buffers
template <typename K>
__device__ void huge_sort_bucket(K* kb, uint32_t* vb, K* ka, uint32_t* va, uint32_t start, uint32_t n, int lo, int hi, K* s_key,
                                 uint32_t* s_val, uint32_t* s_wcnt, uint32_t* s_dstart, uint32_t* s_tmp, uint32_t* s_cnt,
                                 uint32_t* s_base) {
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    constexpr int ITEMS = kTileItems;
    constexpr uint32_t chunk = kTile / kWaves;
    K* sk = kb + start;
    uint32_t* sv = vb + start;
    K* dk = ka + start;
    uint32_t* dv = va + start;
    bool in_b = true;
    for (int shift = lo; shift < hi; shift += 8) {
        const int w = (hi - shift) < 8 ? (hi - shift) : 8;
        const uint32_t mask = (1u << w) - 1u;
        s_cnt[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += kT) atomicAdd(&s_cnt[static_cast<uint32_t>(sk[i] >> shift) & mask], 1u);
        __syncthreads();
        s_base[threadIdx.x] = block_excl_scan256(s_cnt[threadIdx.x], s_tmp);
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += kTile) {
            const uint32_t cnt = (n - c0) < static_cast<uint32_t>(kTile) ? n - c0 : static_cast<uint32_t>(kTile);
            K key[ITEMS];
            uint32_t val[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const uint32_t pos = wave * chunk + r * kWave + lane;
                key[r] = pos < cnt ? sk[c0 + pos] : static_cast<K>(0);
                val[r] = pos < cnt ? sv[c0 + pos] : 0u;
            }
            tile_stage_by_digit<K, ITEMS>(key, val, cnt, chunk, shift, mask, s_key, s_val, s_wcnt, s_dstart, s_tmp);
            for (uint32_t q = threadIdx.x; q < cnt; q += kT) {
                const K kk = s_key[q];
                const uint32_t dg = static_cast<uint32_t>(kk >> shift) & mask;
                const uint32_t o = s_base[dg] + (q - s_dstart[dg]);
                dk[o] = kk;
                dv[o] = s_val[q];
            }
            __syncthreads();
            {   // the digit's running output position moves on by what this chunk held of it
                const int d = threadIdx.x;
                const uint32_t next = d + 1 < kRadix ? s_dstart[d + 1] : cnt;
                s_base[d] += next - s_dstart[d];
            }
            __syncthreads();
        }
        // the pass's stores must be visible to the next pass's loads by the other waves of this workgroup, whose CU may
        // still hold lines of this range in its vector L1 from an earlier pass: agent-scope release + acquire (write-back,
        // L1 invalidate; MI355X_MICROARCH.md, inter-workgroup visibility) -- a few microseconds on a slow path
        __threadfence();
        __syncthreads();
        K* tk = sk; sk = dk; dk = tk;
        uint32_t* tv = sv; sv = dv; dv = tv;
        in_b = !in_b;
    }
    if (!in_b) {   // the sorted run sits in the a buffers: bring it home
        for (uint32_t i = threadIdx.x; i < n; i += kT) {
            dk[i] = sk[i];
            dv[i] = sv[i];
        }
    }
    __syncthreads();
}

// SMALL: one workgroup per (table, digit) bucket, grid T * 256; buckets above the small cap are left to the big launch
template <typename K>
__global__ void __launch_bounds__(kT) seg_local_small_kernel(const SegDesc* desc, const uint32_t* bstart, const uint32_t* bcnt, int mode,
                                                             K* kb, uint32_t* vb) {
    __shared__ K s_key[kSmallCap];
    __shared__ uint32_t s_val[kSmallCap];
    __shared__ uint32_t s_wcnt[kWaves * kRadix];
    __shared__ uint32_t s_dstart[kRadix];
    __shared__ uint32_t s_tmp[kWaves];
    const uint32_t b = blockIdx.x;
    const uint32_t n = bcnt[b];
    if (n < 2 || n > kSmallCap) return;
    int lo, hi;
    local_bits(mode, static_cast<int>(desc[b / kRadix].rbits), lo, hi);
    if (hi <= lo) return;
    local_sort_bucket<K, kSmallItems>(kb, vb, bstart[b], n, lo, hi, s_key, s_val, s_wcnt, s_dstart, s_tmp);
}

// BIG + HUGE: a fixed grid walks the two lists the scan kernel made
template <typename K>
__global__ void __launch_bounds__(kT) seg_local_big_kernel(const SegHeader* hdr, const SegDesc* desc, const uint32_t* bstart,
                                                           const uint32_t* bcnt, const uint32_t* big_list, const uint32_t* huge_list,
                                                           int mode, K* kb, uint32_t* vb, K* ka, uint32_t* va) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    K* s_key = reinterpret_cast<K*>(smem);
    uint32_t* s_val = reinterpret_cast<uint32_t*>(smem + sizeof(K) * kBigCap);
    uint32_t* s_wcnt = s_val + kBigCap;
    uint32_t* s_dstart = s_wcnt + kWaves * kRadix;
    uint32_t* s_tmp = s_dstart + kRadix;
    uint32_t* s_cnt = s_tmp + kWaves;
    uint32_t* s_base = s_cnt + kRadix;
    const uint32_t n_big = hdr->n_big, n_huge = hdr->n_huge;
    // huge buckets first: they are the long poles
    for (uint32_t i = blockIdx.x; i < n_huge; i += gridDim.x) {
        const uint32_t b = huge_list[i];
        int lo, hi;
        local_bits(mode, static_cast<int>(desc[b / kRadix].rbits), lo, hi);
        huge_sort_bucket<K>(kb, vb, ka, va, bstart[b], bcnt[b], lo, hi, s_key, s_val, s_wcnt, s_dstart, s_tmp, s_cnt, s_base);
    }
    for (uint32_t i = blockIdx.x; i < n_big; i += gridDim.x) {
        const uint32_t b = big_list[i];
        int lo, hi;
        local_bits(mode, static_cast<int>(desc[b / kRadix].rbits), lo, hi);
        local_sort_bucket<K, kBigItems>(kb, vb, bstart[b], bcnt[b], lo, hi, s_key, s_val, s_wcnt, s_dstart, s_tmp);
    }
}

inline size_t a256(size_t x) { return (x + 255) / 256 * 256; }
inline size_t tiles_max(size_t n, int T) { return n / kTile + static_cast<size_t>(T) + 1; }

struct Scratch {
    SegHeader* hdr;
    SegDesc* desc;
    uint32_t* tile_seg;
    uint32_t* bh;
    uint32_t* bstart;
    uint32_t* bcnt;
    uint32_t* big_list;
    uint32_t* huge_list;
    size_t total;
};

Scratch scratch_layout(void* base, size_t n, int T) {
    Scratch s;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = p ? p + off : nullptr; off += a256(bytes); return q; };
    const size_t tm = tiles_max(n, T), nb = static_cast<size_t>(T) * kRadix;
    s.hdr = reinterpret_cast<SegHeader*>(take(sizeof(SegHeader)));
    s.desc = reinterpret_cast<SegDesc*>(take(sizeof(SegDesc) * static_cast<size_t>(T)));
    s.tile_seg = reinterpret_cast<uint32_t*>(take(4 * tm));
    s.bh = reinterpret_cast<uint32_t*>(take(4 * tm * kRadix));
    s.bstart = reinterpret_cast<uint32_t*>(take(4 * nb));
    s.bcnt = reinterpret_cast<uint32_t*>(take(4 * nb));
    s.big_list = reinterpret_cast<uint32_t*>(take(4 * nb));
    s.huge_list = reinterpret_cast<uint32_t*>(take(4 * nb));
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
    hipLaunchKernelGGL(seg_prep_tables_kernel, dim3(rq.T), dim3(kT), 0, stream, rq.offsets, rq.idx64, rq.rows, rq.T, rq.B, rq.N,
                       rq.bag_begin, rq.bag_count, rq.weighted ? 1 : 0, s.desc);
    hipLaunchKernelGGL(seg_prep_scan_kernel, dim3(1), dim3(1024), 0, stream, s.desc, rq.T, s.hdr, s.tile_seg);
    if (rq.bag_count > 0) {
        const dim3 gk(static_cast<unsigned>((rq.bag_count + 255) / 256), static_cast<unsigned>(rq.T));
        if (rq.weighted)
            hipLaunchKernelGGL((seg_build_keys_kernel<K, true>), gk, dim3(kT), 0, stream, rq.indices, rq.offsets, rq.idx64, rq.T, rq.B, rq.N,
                               rq.bag_begin, rq.bag_count, s.desc, rq.tshift, keys_a, vals_a, bag_of);
        else
            hipLaunchKernelGGL((seg_build_keys_kernel<K, false>), gk, dim3(kT), 0, stream, rq.indices, rq.offsets, rq.idx64, rq.T, rq.B, rq.N,
                               rq.bag_begin, rq.bag_count, s.desc, rq.tshift, keys_a, vals_a, bag_of);
    }
    // pass 0 reads the request (or the built keys in the a buffers) and writes the b buffers; later passes alternate, so the
    // sorted pairs end in the b buffers iff the pass count is odd (seg_sort_result_in_b).  The local modes run one global
    // pass and then work in place in b, with the a buffers as the huge buckets' spare space.
    const int total = seg_sort_passes(mode, rq.rbits_max);
    for (int p = 0; p < total; ++p) {
        PassSrc<K> src;
        src.indices = rq.indices;
        src.idx64 = rq.idx64;
        src.first = p == 0 ? 1 : 0;
        src.tshift = rq.tshift;
        src.bag_begin = static_cast<uint32_t>(rq.bag_begin);
        src.keys = p == 0 ? keys_a : (p % 2 == 1 ? keys_b : keys_a);
        src.vals = p == 0 ? vals_a : (p % 2 == 1 ? vals_b : vals_a);
        K* kout = (p % 2 == 0) ? keys_b : keys_a;
        uint32_t* vout = (p % 2 == 0) ? vals_b : vals_a;
        hipLaunchKernelGGL((seg_hist_kernel<K>), dim3(tm), dim3(kT), 0, stream, s.hdr, s.desc, s.tile_seg, src, mode, p, s.bh);
        hipLaunchKernelGGL(seg_scan_kernel, dim3(rq.T), dim3(kRadix), 0, stream, s.hdr, s.desc, s.bh, s.bstart, s.bcnt,
                           (mode != 0 && p == 0) ? 1 : 0, mode, s.big_list, s.huge_list);
        hipLaunchKernelGGL((seg_scatter_kernel<K>), dim3(tm), dim3(kT), 0, stream, s.hdr, s.desc, s.tile_seg, src, mode, p, s.bh, s.bstart,
                           kout, vout);
    }
    if (mode != 0) {
        hipLaunchKernelGGL((seg_local_small_kernel<K>), dim3(static_cast<unsigned>(rq.T) * kRadix), dim3(kT), 0, stream, s.desc, s.bstart,
                           s.bcnt, mode, keys_b, vals_b);
        const size_t lds = sizeof(K) * kBigCap + 4 * (kBigCap + kWaves * kRadix + kRadix + kWaves + kRadix + kRadix);
        static bool attr_set[2] = {false, false};
        if (!attr_set[sizeof(K) == 8]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_local_big_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds));
            attr_set[sizeof(K) == 8] = true;
        }
        hipLaunchKernelGGL((seg_local_big_kernel<K>), dim3(256), dim3(kT), lds, stream, s.hdr, s.desc, s.bstart, s.bcnt, s.big_list,
                           s.huge_list, mode, keys_b, vals_b, keys_a, vals_a);
    }
    return hipGetLastError();
}

template hipError_t seg_sort_pairs<uint32_t>(const SegSortRequest&, int, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, void*,
                                             hipStream_t);
template hipError_t seg_sort_pairs<uint64_t>(const SegSortRequest&, int, uint64_t*, uint64_t*, uint32_t*, uint32_t*, uint32_t*, void*,
                                             hipStream_t);

}  // namespace pm
