// param_amd/csrc/radix_sort.hip -- stable LSD radix sort of (key, uint32 value) pairs for gfx950, written for the
// sorted EmbeddingBag backward (embbag_bwd_sorted.hip): 8-bit digits, three plain kernels per pass and NO communication
// between workgroups inside a kernel (kernel boundaries are the only synchronisation), so there is nothing to get wrong
// about memory ordering across the eight XCDs' L2s:
//
//   rs_hist_kernel     per 4096-element tile: how many keys carry each digit value             -> bh[tile][256]
//   rs_scan_kernel     per digit: exclusive prefix of the tile counts over tiles (in place)     -> bh, total[256]
//                      (rs_scan_seg_kernel: per segment, when the array is a sequence of independently sorted segments)
//   rs_scatter_kernel  per tile: stable rank of every element among the tile's elements of the same digit (wave-level
//                      match by eight ballots, per-wave running counters in LDS, waves own consecutive quarters of the
//                      tile), tile reordered by digit in LDS, then written out so that elements of one digit leave as
//                      one contiguous run (64-128 B segments instead of single 4-byte scatters)
//
// The element count may live in DEVICE memory (d_count): every kernel reads it and tiles past the end exit, so a caller
// can sort "however many elements a previous kernel produced" without a host round trip (the hybrid backward compacts
// the duplicate lookups on the device).  The 63 MB of pairs of the benchmark step (7.86 M lookups) stay in the 256 MB
// memory-side cache across passes; HBM is not what bounds this.
#include <cstdlib>

#include "common.h"

namespace pm {
namespace {

constexpr int kRsThreads = 256;
constexpr int kRsWaves = kRsThreads / kWave;           // 4
constexpr int kRsItems = 16;                           // elements per thread
constexpr int kRsTile = kRsThreads * kRsItems;         // 4096 elements per workgroup
constexpr int kRsWaveChunk = kRsTile / kRsWaves;       // 1024 consecutive tile positions per wave
constexpr int kRsRadix = 256;
constexpr int kScanThreads = 1024;
constexpr int kScanDigits = 16;                        // digits per scan workgroup
constexpr int kScanChunks = kScanThreads / kScanDigits;  // 64 row chunks

__device__ __forceinline__ uint32_t rs_count(const uint32_t* d_count, uint32_t n_max) {
    if (!d_count) return n_max;
    const uint32_t c = *d_count;
    return c < n_max ? c : n_max;
}

// lanes of the wave (among those with `valid`) whose 8-bit digit equals this lane's
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// FROM_IDX (first pass of the sorted backward's per-table sort, fixed pooling): the keys are not read from memory but
// formed from the request's index array -- the digit of the first pass is the low bits of the row id --, which saves
// the key-building kernel and its 126 MB of traffic (RsSource below).
template <typename K, bool FROM_IDX>
__global__ void __launch_bounds__(kRsThreads) rs_hist_kernel(const K* keys, const uint32_t* d_count, uint32_t n_max,
                                                             int shift, uint32_t mask, uint32_t* bh, const RsSource src) {
    __shared__ uint32_t h[kRsRadix];
    const uint32_t n = rs_count(d_count, n_max);
    const uint64_t tile0 = static_cast<uint64_t>(blockIdx.x) * kRsTile;
    if (tile0 >= n) return;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x % kWave;
#pragma unroll 4
    for (int k = 0; k < kRsItems; ++k) {
        const uint64_t i = tile0 + static_cast<uint64_t>(k) * kRsThreads + threadIdx.x;
        const bool valid = i < n;
        uint32_t d = 0u;
        if (valid) d = FROM_IDX ? (static_cast<uint32_t>(load_index(src.indices, static_cast<int64_t>(i), src.idx64)) >> shift) & mask
                                : static_cast<uint32_t>(keys[i] >> shift) & mask;
        // A wave whose 64 keys share the digit (the top digits of a Zipf head, of small tables, of any nearly sorted
        // input) adds once; otherwise one LDS atomic per lane -- distinct digits do not conflict, and the eight-ballot
        // match that would merge the equal ones costs more than the conflicts of a mixed wave (24.8 -> ~10 us per pass).
        const uint64_t vmask = __ballot(valid);
        const uint32_t first = __builtin_amdgcn_readfirstlane(d);
        const bool uniform = __ballot(valid && d != first) == 0 && (vmask & 1ull);
        if (uniform) {
            if (lane == 0) atomicAdd(&h[first], static_cast<uint32_t>(__popcll(vmask)));
        } else if (valid) {
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    bh[static_cast<uint64_t>(blockIdx.x) * kRsRadix + threadIdx.x] = h[threadIdx.x];
}

// bh[tile][d] -> exclusive prefix over tiles (in place), total[d] = sum over tiles.  One workgroup per 16 digits.
__global__ void __launch_bounds__(kScanThreads) rs_scan_kernel(uint32_t* bh, uint32_t* total, const uint32_t* d_count,
                                                               uint32_t n_max) {
    __shared__ uint32_t s_sum[kScanChunks][kScanDigits];
    const uint32_t n = rs_count(d_count, n_max);
    const uint32_t nb = (n + kRsTile - 1) / kRsTile;
    const int dl = threadIdx.x % kScanDigits;
    const int c = threadIdx.x / kScanDigits;
    const int d = blockIdx.x * kScanDigits + dl;
    const uint32_t per = (nb + kScanChunks - 1) / kScanChunks;
    const uint32_t r0 = static_cast<uint32_t>(c) * per < nb ? static_cast<uint32_t>(c) * per : nb;
    const uint32_t r1 = r0 + per < nb ? r0 + per : nb;
    uint32_t sum = 0;
#pragma unroll 8
    for (uint32_t r = r0; r < r1; ++r) sum += bh[static_cast<uint64_t>(r) * kRsRadix + d];
    s_sum[c][dl] = sum;
    __syncthreads();
    uint32_t base = 0;
    for (int cc = 0; cc < c; ++cc) base += s_sum[cc][dl];
    if (c == kScanChunks - 1) total[d] = base + sum;
    // in-place rewrite, eight rows per round trip (loads of a batch are issued before its stores: a load-store-load
    // chain through the same array would serialise on the L2 latency)
    uint32_t run = base;
    for (uint32_t r = r0; r < r1; r += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (r + u < r1) ? bh[static_cast<uint64_t>(r + u) * kRsRadix + d] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (r + u < r1) bh[static_cast<uint64_t>(r + u) * kRsRadix + d] = run;
            run += v[u];
        }
    }
}

// Segmented form (every segment = seg_tiles whole tiles, sorted independently): one workgroup per segment, thread = digit,
// rows of the segment walked in order (each row is one coalesced 1 KB read), eight per round trip.
__global__ void __launch_bounds__(kRsRadix) rs_scan_seg_kernel(uint32_t* bh, uint32_t* total, uint32_t seg_tiles) {
    const int d = threadIdx.x;
    const uint64_t r0 = static_cast<uint64_t>(blockIdx.x) * seg_tiles;
    uint32_t run = 0;
    for (uint32_t r = 0; r < seg_tiles; r += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (r + u < seg_tiles) ? bh[(r0 + r + u) * kRsRadix + d] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (r + u < seg_tiles) bh[(r0 + r + u) * kRsRadix + d] = run;
            run += v[u];
        }
    }
    total[static_cast<uint64_t>(blockIdx.x) * kRsRadix + d] = run;
}

// exclusive scan of one value per thread over the 256 threads of the workgroup; s_tmp: kRsWaves words
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_tmp) {
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
    __syncthreads();   // s_tmp may be reused by the next call
    return base + incl - v;
}

template <typename K, bool FROM_IDX>
__global__ void __launch_bounds__(kRsThreads) rs_scatter_kernel(const K* kin, const uint32_t* vin, K* kout, uint32_t* vout,
                                                                const uint32_t* d_count, uint32_t n_max, int shift,
                                                                uint32_t mask, const uint32_t* prefix, const uint32_t* total,
                                                                uint32_t seg_tiles, const RsSource src) {
    __shared__ K s_key[kRsTile];
    __shared__ uint32_t s_val[kRsTile];
    __shared__ uint32_t s_wcnt[kRsWaves][kRsRadix];   // per wave: running digit counts, later the wave's base inside the digit
    __shared__ uint32_t s_dstart[kRsRadix];           // first tile-local position of digit d after the reorder
    __shared__ uint32_t s_gbase[kRsRadix];            // global position of the tile's first element of digit d
    __shared__ uint32_t s_tmp[kRsWaves];
    const uint32_t n = rs_count(d_count, n_max);
    const uint64_t tile0 = static_cast<uint64_t>(blockIdx.x) * kRsTile;
    if (tile0 >= n) return;
    const uint32_t cnt = (n - tile0) < kRsTile ? static_cast<uint32_t>(n - tile0) : kRsTile;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    for (int i = threadIdx.x; i < kRsWaves * kRsRadix; i += kRsThreads) (&s_wcnt[0][0])[i] = 0;

    // wave w owns tile positions [w*1024, (w+1)*1024), taken 64 at a time in position order
    K key[kRsItems];
    uint32_t val[kRsItems];
    uint32_t rank[kRsItems];
#pragma unroll
    for (int r = 0; r < kRsItems; ++r) {
        const uint32_t pos = wave * kRsWaveChunk + r * kWave + lane;
        const bool valid = pos < cnt;
        if (FROM_IDX) {
            // key = (segment = table) << tshift | row; value = the lookup's bag inside its table (segments are whole tables of
            // B bags with `pooling` lookups each)
            const uint32_t seg = blockIdx.x / seg_tiles;
            const uint32_t in_seg = (blockIdx.x % seg_tiles) * static_cast<uint32_t>(kRsTile) + pos;
            key[r] = valid ? ((static_cast<K>(seg) << src.tshift) |
                              static_cast<K>(load_index(src.indices, static_cast<int64_t>(tile0 + pos), src.idx64))) : static_cast<K>(0);
            val[r] = valid ? in_seg / src.pooling : 0u;
        } else {
            key[r] = valid ? kin[tile0 + pos] : static_cast<K>(0);
            val[r] = valid ? vin[tile0 + pos] : 0u;
        }
    }
    __syncthreads();   // counters zeroed
    uint32_t* wcnt = s_wcnt[wave];
#pragma unroll
    for (int r = 0; r < kRsItems; ++r) {
        const uint32_t pos = wave * kRsWaveChunk + r * kWave + lane;
        const bool valid = pos < cnt;
        const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & mask;
        const uint64_t m = match_digit(d, valid);
        const uint32_t below = static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
        const uint32_t base = valid ? wcnt[d] : 0u;
        rank[r] = base + below;
        // the lowest lane of each match set advances the digit's counter; one wave executes its LDS operations in
        // program order, so the next round's reads see it (the rounds of one wave are sequential by construction)
        if (valid && below == 0) wcnt[d] = base + static_cast<uint32_t>(__popcll(m));
    }
    __syncthreads();
    {
        const int d = threadIdx.x;   // 256 threads = 256 digits
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < kRsWaves; ++w) {
            const uint32_t c = s_wcnt[w][d];
            s_wcnt[w][d] = acc;      // wave w's elements of digit d start this far into the digit's tile-local run
            acc += c;
        }
        const uint32_t dstart = block_excl_scan(acc, s_tmp);
        // segmented: the tile's segment starts at seg * seg_tiles * tile size and has its own digit totals
        const uint32_t seg = seg_tiles ? blockIdx.x / seg_tiles : 0u;
        const uint32_t gdigit = block_excl_scan(total[static_cast<uint64_t>(seg) * kRsRadix + d], s_tmp);
        s_dstart[d] = dstart;
        s_gbase[d] = seg * seg_tiles * static_cast<uint32_t>(kRsTile) + gdigit + prefix[static_cast<uint64_t>(blockIdx.x) * kRsRadix + d];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRsItems; ++r) {
        const uint32_t pos = wave * kRsWaveChunk + r * kWave + lane;
        if (pos < cnt) {
            const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & mask;
            const uint32_t q = s_dstart[d] + wcnt[d] + rank[r];
            s_key[q] = key[r];
            s_val[q] = val[r];
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < kRsItems; ++k) {
        const uint32_t q = k * kRsThreads + threadIdx.x;
        if (q < cnt) {
            const K kk = s_key[q];
            const uint32_t d = static_cast<uint32_t>(kk >> shift) & mask;
            const uint64_t g = static_cast<uint64_t>(s_gbase[d]) + (q - s_dstart[d]);
            kout[g] = kk;
            vout[g] = s_val[q];
        }
    }
}

inline size_t rs_tiles(size_t n_max) { return (n_max + kRsTile - 1) / kRsTile; }

}  // namespace

// tile counts + per-segment digit totals (at most one segment per tile)
size_t rs_scratch_bytes(size_t n_max) {
    return (2 * rs_tiles(n_max) * kRsRadix + kRsRadix) * sizeof(uint32_t) + 256;
}

int rs_num_passes(int begin_bit, int end_bit) {
    const int bits = end_bit - begin_bit;
    return bits <= 0 ? 0 : (bits + 7) / 8;
}

// Sorts by key bits [begin_bit, end_bit).  The pairs start in (keys_a, vals_a); passes alternate between the a and b
// buffers, so the result is in the b buffers iff rs_num_passes() is odd (0 passes: nothing moves, result in a).
// seg_len > 0: the array is a sequence of segments of seg_len elements (a multiple of the 4096-element tile, n_max a
// multiple of seg_len, no d_count), each sorted on its own -- the backward's tables (x bag phases) when every bag has the
// same number of lookups: the table bits then need no pass at all.
// src != nullptr (segments only, begin_bit 0): the FIRST pass forms its pairs from the request's index array instead of
// reading (keys_a, vals_a), which are then never touched -- key = segment << tshift | index, value = position in the
// segment / pooling (the bag).
template <typename K>
hipError_t rs_sort_pairs(K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n_max, const uint32_t* d_count,
                         int begin_bit, int end_bit, void* scratch, hipStream_t stream, size_t seg_len, const RsSource* src) {
    if (n_max == 0) return hipSuccess;
    if (n_max > 0xffffffffull) return hipErrorInvalidValue;
    if (seg_len && (seg_len % kRsTile || n_max % seg_len || d_count)) return hipErrorInvalidValue;
    if (src && (!seg_len || begin_bit != 0 || !src->indices || src->pooling == 0 || rs_num_passes(begin_bit, end_bit) == 0))
        return hipErrorInvalidValue;
    const RsSource none{nullptr, 0, 0, 1};
    const uint32_t seg_tiles = static_cast<uint32_t>(seg_len / kRsTile);
    const int passes = rs_num_passes(begin_bit, end_bit);
    const int bits = end_bit - begin_bit;
    uint32_t* bh = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* total = bh + rs_tiles(n_max) * kRsRadix;
    const unsigned grid = static_cast<unsigned>(rs_tiles(n_max));
    const uint32_t n32 = static_cast<uint32_t>(n_max);
    // digit widths as even as possible: 30 bits -> 8, 8, 7, 7
    int sh[16], wd[16];
    for (int p = 0, s = begin_bit; p < passes; ++p) {
        wd[p] = (bits - (s - begin_bit) + (passes - p) - 1) / (passes - p);
        sh[p] = s;
        s += wd[p];
    }
    // (round 3 measured what the ascending-row order is worth to the apply's address translation with the lowest digit sorted
    // LAST -- equal keys adjacent, neighbours in the output no longer neighbours in key space: HISTORY; the switch is gone)
    for (int p = 0; p < passes; ++p) {
        const int shift = sh[p], w = wd[p];
        const uint32_t mask = (1u << w) - 1u;
        const K* kin = (p % 2 == 0) ? keys_a : keys_b;
        K* kout = (p % 2 == 0) ? keys_b : keys_a;
        const uint32_t* vin = (p % 2 == 0) ? vals_a : vals_b;
        uint32_t* vout = (p % 2 == 0) ? vals_b : vals_a;
        const bool first_from_idx = src != nullptr && p == 0;
        if (first_from_idx)
            hipLaunchKernelGGL((rs_hist_kernel<K, true>), dim3(grid), dim3(kRsThreads), 0, stream, kin, d_count, n32, shift, mask, bh, *src);
        else
            hipLaunchKernelGGL((rs_hist_kernel<K, false>), dim3(grid), dim3(kRsThreads), 0, stream, kin, d_count, n32, shift, mask, bh, none);
        if (seg_tiles)
            hipLaunchKernelGGL(rs_scan_seg_kernel, dim3(grid / seg_tiles), dim3(kRsRadix), 0, stream, bh, total, seg_tiles);
        else
            hipLaunchKernelGGL(rs_scan_kernel, dim3(kRsRadix / kScanDigits), dim3(kScanThreads), 0, stream, bh, total, d_count, n32);
        if (first_from_idx)
            hipLaunchKernelGGL((rs_scatter_kernel<K, true>), dim3(grid), dim3(kRsThreads), 0, stream, kin, vin, kout, vout, d_count, n32,
                               shift, mask, bh, total, seg_tiles, *src);
        else
            hipLaunchKernelGGL((rs_scatter_kernel<K, false>), dim3(grid), dim3(kRsThreads), 0, stream, kin, vin, kout, vout, d_count, n32,
                               shift, mask, bh, total, seg_tiles, none);
    }
    return hipGetLastError();
}

template hipError_t rs_sort_pairs<uint32_t>(uint32_t*, uint32_t*, uint32_t*, uint32_t*, size_t, const uint32_t*, int, int,
                                            void*, hipStream_t, size_t, const RsSource*);
template hipError_t rs_sort_pairs<uint64_t>(uint64_t*, uint64_t*, uint32_t*, uint32_t*, size_t, const uint32_t*, int, int,
                                            void*, hipStream_t, size_t, const RsSource*);

}  // namespace pm
