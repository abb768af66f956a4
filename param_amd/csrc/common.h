// param_amd/csrc/common.h -- shared device/host declarations for libparam_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "param_amd.h"

namespace pm {

constexpr int kBlock = 256;  // 4 wave64 per workgroup
constexpr int kWave = 64;
constexpr int kXcds = 8;     // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only)

// native clang vector types (the nontemporal builtins reject HIP_vector_type wrappers)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Table pointers reach the kernels through a device array (tables[t]) or LDS copies of it, so the compiler cannot
// tell they point to global memory and emits FLAT loads/stores/atomics for the rows.  FLAT instructions count against
// lgkmcnt as well as vmcnt: every wait for an LDS read (the staged indices) then also waits for all row loads in flight.
// The row accesses therefore go through explicit global-address-space pointers (global_load_dwordx4 ...).
#define PM_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const PM_GLOBAL T* as_global(const void* p) { return (const PM_GLOBAL T*)(p); }
template <typename T>
__device__ __forceinline__ PM_GLOBAL T* as_global(void* p) { return (PM_GLOBAL T*)(p); }

// Kernel-argument block shared by forward / backward / check (passed by value: SGPRs).
struct KParams {
    const void* const* tables;   // fwd: source tables; bwd: destination tables
    const int64_t* rows;
    const int32_t* dims;
    const int64_t* out_offsets;
    const void* indices;
    const void* offsets;
    const float* psw;
    float* io;                   // fwd: out ; bwd: grad (read-only)
    int64_t out_stride;
    int64_t B;                   // bags per table
    int64_t N;                   // total indices
    int64_t bag_begin;
    int64_t bag_count;
    int32_t T;
    int32_t tiles_per_table;     // ceil(bag_count / bags_per_block)
    int32_t bags_per_block;
    int32_t idx_cap;             // LDS index-tile capacity (entries)
    int32_t idx64;               // 1: int64 indices/offsets, 0: int32
    int32_t xcd_affine;          // 1: table t is served by XCD t % 8 (requires T % 8 == 0); 2: tile-major block order (t = b % T);
                                 // 3: XCD x serves the x-th eighth of the table-major tile order (any T)
    int32_t nt_loads;            // 1: non-temporal table-row loads
    int32_t ordered;             // forward: 1 = ragged request, lane groups take the longest bags of a tile first
    int32_t stage_out;           // forward: > 0 = collect the tile's pooled rows in LDS (this many floats per row), write at tile end
    int32_t stage_bags;          // forward: rows the staging buffer holds (tiles with more bags write row by row)
    int32_t out_bits;            // forward: 0 = fp32 output; 16 / 8 / 4 / 2 = io holds row-wise quantised rows (rowquant.hip), staged only
    int32_t flat_bags;           // forward: > 0 = the flat-walk kernel (short-bag requests): bags per tile derived per table on the
                                 // device, at most this many (= bags_per_block, which sizes the LDS offsets array)
    int32_t flat_target;         // ... lookups per tile aimed at
    int32_t flat_compact;        // ... > 0 (round 6): the launch is this many RESIDENT workgroups that each establish every table's tile
                                 // count from the offsets (LDS prefix) and walk the table-major tile order b, b + grid, ...: no workgroup
                                 // is dispatched for a tile that does not exist.  0: grid = T x tiles_per_table, surplus workgroups leave
    int32_t gblk_shift;          // backward kernels: blocked gradient layout (ABI v6, pm_embbag_batch::grad_block_shift): bag b of table t at
    int64_t gblk_extra;          //   io + out_offsets[t] + b * out_stride + (b >> gblk_shift) * gblk_extra ; no blocking: extra = 0
    float alpha;                 // bwd scale
};

// element offset of bag `bag`'s gradient row inside its table's slice (blocked layouts add a per-block term; extra == 0 otherwise)
// (`extra` is a kernel argument: the test is a scalar branch, and the un-blocked layouts -- every benchmark line -- skip the second
// 64-bit multiply; computed unconditionally it cost the Zipf sorted apply 1.1 %, same-box A/B)
__device__ __forceinline__ int64_t grad_bag_offset(int64_t bag, int64_t out_stride, int shift, int64_t extra) {
    int64_t o = bag * out_stride;
    if (__builtin_expect(extra != 0, 0)) o += (bag >> shift) * extra;
    return o;
}

__device__ __forceinline__ int64_t load_index(const void* p, int64_t i, int idx64) {
    return idx64 ? as_global<int64_t>(p)[i] : static_cast<int64_t>(as_global<int32_t>(p)[i]);
}

// End of global bag g (g in [0, T*B)): next offset, or N for the very last bag
// (include_last_offset=False rule; a trailing offsets[T*B] entry is never read).
__device__ __forceinline__ int64_t bag_start_or_end(const KParams& p, int64_t g) {
    const int64_t TB = static_cast<int64_t>(p.T) * p.B;
    return (g < TB) ? load_index(p.offsets, g, p.idx64) : p.N;
}

// blockIdx -> (table, bag tile).  With xcd_affine the 8 XCDs each own the tables
// t == xcd (mod 8), so one table's hot rows live in exactly one XCD's 4 MiB L2
// instead of being replicated in all eight (placement is a speed matter only).
// XCD-contiguous order for any table count (xcd_affine == 3, round 4): XCD x (blocks are dispatched round-robin over the 8
// XCDs) serves the x-th eighth of the table-major tile order -- a bijection of [0, total) for any total, so the grid stays
// T x tiles_per_table.  The tiles an XCD has in flight then belong to one or two tables, like t % 8 gives for T % 8 == 0.
__device__ __forceinline__ int xcd_contiguous_tile(int bid, int total) {
    const int q = total / kXcds, r = total % kXcds, x = bid % kXcds;
    return x * q + (x < r ? x : r) + bid / kXcds;
}

__device__ __forceinline__ void block_to_tile(const KParams& p, int& t, int& tile, int bid = blockIdx.x) {
    if (p.xcd_affine == 3) {
        const int g = xcd_contiguous_tile(bid, p.T * p.tiles_per_table);
        t = g / p.tiles_per_table;
        tile = g % p.tiles_per_table;
    } else if (p.xcd_affine == 2) {
        // tile-major: consecutive blocks serve the same tile index of consecutive tables, so every table advances at the same
        // rate.  For requests whose tables have very different pooling factors (Criteo multi-hot 1 .. 100): in table-major
        // order the heaviest table's workgroups -- each a chain of 25 round trips -- are dispatched together, late, and the
        // launch ends in their tail; interleaved, they start throughout the launch.
        t = bid % p.T;
        tile = bid / p.T;
    } else if (p.xcd_affine) {
        const int xcd = bid % kXcds;
        const int slot = bid / kXcds;
        t = xcd + kXcds * (slot / p.tiles_per_table);
        tile = slot % p.tiles_per_table;
    } else {
        t = bid / p.tiles_per_table;
        tile = bid % p.tiles_per_table;
    }
}

// Stage the tile's offsets (absolute, int64) and -- when the tile's index range fits --
// its indices (narrowed to int32: rows[t] < 2^31 is the caller's contract, verified by pm_embbag_check and
// by the Python modules) and per-sample
// weights into LDS with coalesced loads.  Returns true if indices were staged.
// LDS layout: int64 s_off[bags_per_block + 1] | int32 s_idx[idx_cap] | float s_w[idx_cap]
template <bool WEIGHTED>
__device__ __forceinline__ bool stage_tile_at(const KParams& p, int t, int64_t bag0, int nb, char* smem,
                                              int64_t*& s_off, int32_t*& s_idx, float*& s_w);

template <bool WEIGHTED>
__device__ __forceinline__ bool stage_tile(const KParams& p, int t, int tile, char* smem, int& nb,
                                           int64_t*& s_off, int32_t*& s_idx, float*& s_w) {
    const int64_t bag0 = p.bag_begin + static_cast<int64_t>(tile) * p.bags_per_block;
    const int64_t left = p.bag_begin + p.bag_count - bag0;
    nb = left < p.bags_per_block ? static_cast<int>(left) : p.bags_per_block;
    return stage_tile_at<WEIGHTED>(p, t, bag0, nb, smem, s_off, s_idx, s_w);
}

// the same for a tile given by its first bag and bag count (work tiles)
template <bool WEIGHTED>
__device__ __forceinline__ bool stage_tile_at(const KParams& p, int t, int64_t bag0, int nb, char* smem,
                                              int64_t*& s_off, int32_t*& s_idx, float*& s_w) {
    const int64_t g0 = static_cast<int64_t>(t) * p.B + bag0;

    s_off = reinterpret_cast<int64_t*>(smem);
    s_idx = reinterpret_cast<int32_t*>(smem + (static_cast<size_t>(p.bags_per_block + 2) / 2 * 2) * sizeof(int64_t));
    s_w = reinterpret_cast<float*>(s_idx + p.idx_cap);

    for (int i = threadIdx.x; i <= nb; i += kBlock) s_off[i] = bag_start_or_end(p, g0 + i);
    __syncthreads();
    const int64_t base = s_off[0];
    const int64_t cnt = s_off[nb] - base;
    const bool staged = cnt <= p.idx_cap;
    if (staged) {
        for (int i = threadIdx.x; i < static_cast<int>(cnt); i += kBlock) {
            s_idx[i] = static_cast<int32_t>(load_index(p.indices, base + i, p.idx64));
            if (WEIGHTED) s_w[i] = as_global<float>(p.psw)[base + i];
        }
        __syncthreads();
    }
    return staged;
}

__host__ __device__ inline size_t tile_lds_bytes(int bags_per_block, int idx_cap, bool weighted) {
    return (static_cast<size_t>(bags_per_block + 2) / 2 * 2) * sizeof(int64_t) +
           static_cast<size_t>(idx_cap) * 4 * (weighted ? 2 : 1);
}

// ---- host-side launchers implemented in the kernel files ----------------------------------
hipError_t launch_embbag_fwd(const KParams& p, int weight_dtype, int max_dim, int unroll,
                             hipStream_t stream);
hipError_t launch_embbag_fwd_split(const KParams& p, int weight_dtype, int max_dim, hipStream_t stream);
#ifdef PM_ALTERNATES
hipError_t launch_embbag_bwd(const KParams& p, int dst_dtype, int max_dim, hipStream_t stream);   // the atomic backward (embbag_bwd.hip)
#endif
hipError_t launch_embbag_check(const KParams& p, int32_t* d_err, int vec, int max_dim, int64_t fixed_pooling,
                               int uniform_dims, hipStream_t stream);
hipError_t launch_fill_random(void* dst, int64_t count, int dtype, int dist, float lo, float hi,
                              uint64_t seed, hipStream_t stream);

// sort-based deterministic backward (embbag_bwd_sorted.hip)
hipError_t sorted_workspace_bytes(const KParams& p, int64_t max_rows, int max_dim, size_t& bytes);
// defer_ok: the caller issues the apply of this very request right behind the sort (the fused entry points): only then may the
// sort stop after its first part (segments, verdicts, dup maps) and leave the rest to the apply call (hybrid backward), which
// reads the request's indices again.  A sort issued on its own always consumes the request completely.
hipError_t sort_indices(const KParams& p, int64_t max_rows, int max_dim, int64_t fixed_pooling, int phases, void* workspace,
                        hipStream_t stream, bool defer_ok = false);
int bwd_sorted_plan_check(const KParams& p, int64_t max_rows, const void* workspace, bool adagrad);
int sorted_pairs_info(const KParams& p, int64_t max_rows, int max_dim, const void* workspace, const void** keys, const uint32_t** vals,
                      const uint32_t** d_count, int* key_bytes, int* tshift);
std::string sort_plan_describe(const KParams& p, int64_t max_rows, int64_t fixed_pooling, int phases);
hipError_t bwd_sorted_apply(const KParams& p, int64_t max_rows, int dst_dtype, int max_dim, const void* workspace,
                            float* const* momentum, const pm_rowwise_adagrad* opt, hipStream_t stream);

#ifdef PM_ALTERNATES
// own stable LSD radix sort of (key, uint32) pairs (radix_sort.hip); element count optionally read from device memory
size_t rs_scratch_bytes(size_t n_max);
int rs_num_passes(int begin_bit, int end_bit);   // result lands in the b buffers iff odd
struct RsSource {            // first-pass input formed on the fly (sorted backward, per-table segments, fixed pooling)
    const void* indices;     // the request's index array (int64 or int32)
    int idx64;
    int tshift;              // key = segment << tshift | index
    uint32_t pooling;        // value = position inside the segment / pooling
};
template <typename K>
hipError_t rs_sort_pairs(K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n_max, const uint32_t* d_count,
                         int begin_bit, int end_bit, void* scratch, hipStream_t stream, size_t seg_len = 0,
                         const RsSource* src = nullptr);
#endif

// Hybrid backward (round 4): tables whose lookups are (nearly) all to distinct rows skip the sort.  The sort's first kernel
// classifies every table on the device; a HYBRID table's lookups are tested against four hashed "this row was looked up twice"
// bitmaps (Bloom-style: built in LDS by hyb_mark_kernel, no false negatives); rows looked up once are read-modify-written
// bag-major with the bag's gradient slice in registers (bwd_unique_kernel), only the flagged lookups are compacted into the
// sort's input.  Everything else about the sort and the sorted apply is unchanged.
constexpr int kHybMaxTables = 128;              // tables eligible for the hybrid path (256 KB of bitmaps each)
// The dup test is a BLOCKED Bloom filter: a row owns ONE 32-bit word of its table's map (multiplicative hash) and up to four
// bit positions inside it (second hash), so testing a lookup is one 4-byte gather and marking it one LDS atomic.  Measured
// against four independent 2^19-bit maps (round 4, visit 3): the emit kernel's four gathers per lookup were 125 us of
// address processing for the benchmark request.  Simulated at 163 840 uniform lookups into 10 M rows: 2.95 % of the lookups
// flagged (1.65 % true repeats + 1.3 % false), 1.9 % with twice the words; no false negatives by construction.
#ifndef PM_BLOOM_K
#define PM_BLOOM_K 4
#endif
constexpr int kBloomK = PM_BLOOM_K;             // slices of a table's map = mark workgroups per table (-DPM_BLOOM_K=8: experiment builds)
constexpr int kBloomWords = (1 << 16) / kBloomK;   // 32-bit words per slice (4 slices: 64 KB each, seen + dup of a slice fill 128 KB of LDS)
constexpr int kBloomTableWords = kBloomK * kBloomWords;   // 65 536 words = 256 KB per table
constexpr uint32_t kHybMaxCount = 1u << 18;     // lookups per table a map of kBloomK slices separates (beyond: a 2^21-bit map flags too many unique rows)
// Round 5: the map GROWS with the table's lookups -- a rank of an N-GPU table-wise sharded step serves the GLOBAL batch for its
// tables (N x 8192 bags: 327 K / 655 K / 1.3 M lookups per table at N = 2 / 4 / 8), and at 2^18 the hybrid path was an N = 1
// optimisation only.  Slices per table (= mark workgroups per table, each rescanning the table's lookups): kBloomK up to 2^18
// lookups, doubling with the count up to kBloomKMax; the number is a function of the request's sizes alone (N / T), so the
// workspace query, the sort and the apply agree on it.
constexpr int kBloomKMax = 64;
__host__ __device__ inline int hyb_slices(int64_t n_lookups, int T) {
    const int64_t per = T > 0 ? (n_lookups + T - 1) / T : 0;
    int k = PM_BLOOM_K;
    while (k < kBloomKMax && per > static_cast<int64_t>(k) * (kHybMaxCount / PM_BLOOM_K)) k *= 2;
    return k;
}
constexpr uint32_t kHybMinCount = 8192;         // ... and below which a table is not worth three extra kernels
constexpr int kUniqueBags = 128;                // bags per tile of the bag-major apply: ONE value for the sort-time guard, the apply's launch and
                                                // the compaction of its per-tile lists (no knob: the three must agree)
constexpr int kCompactMaxTiles = 4096;          // bag-major tiles per table the compaction scans (sort-time guard: bags <= 128 x 4096)
bool seg_sort_hybrid_available();               // the mark kernel's 128 KB of dynamic LDS can be had on this device (probed once)
struct HybTable {            // one per table (device): written by the sort's first kernel and by the compaction
    uint32_t mode;           // 0: every lookup through the sort; 1: hybrid.  Final verdict, written by hyb_mark_kernel's first workgroup
    uint32_t pooling;        // the table's pooling factor (hybrid tables have one)
    uint32_t n_dup;          // lookups left to the sort (hybrid tables: written by hyb_rest_kernel; 0 for a table it finished in LDS)
    uint32_t cand;           // the table qualifies on its own (seg_prep_tables_kernel); it goes hybrid if the qualifying tables
                             // together hold at least half of the request's lookups -- the mark + bag-major kernels have fixed
                             // costs that a few small tables do not repay (Criteo uniform: 5 of 26 tables, 24 % of the lookups,
                             // 0.49 -> 0.57 ms)
};
// word of a row in a map of 2^wbits words (wbits = 14 + log2 slices: 16 for the four slices of a table of up to 2^18 lookups)
__host__ __device__ inline uint32_t bloom_word(uint32_t row, int wbits) { return (row * 0x9E3779B1u) >> (32 - wbits); }
__host__ __device__ inline int bloom_wbits(int slices) { int b = 14; while ((1 << (b - 14)) < slices) ++b; return b; }
__host__ __device__ inline uint32_t bloom_mask(uint32_t row) {
    const uint32_t h = row * 0x85EBCA77u;
    return (1u << (h >> 27)) | (1u << ((h >> 22) & 31u)) | (1u << ((h >> 17) & 31u)) | (1u << ((h >> 12) & 31u));
}
struct HybArgs {             // hybrid part of a sort request
    int allow;               // 0: every table is classified "sort" and nothing else of the hybrid path runs; 1: on; 2: structural eligibility only (tests)
    int slices;              // map slices per table (hyb_slices(N, T): a power of two, kBloomK .. kBloomKMax)
};
// (the per-table records, the dup bitmaps [min(T, kHybMaxTables)][kBloomTableWords] and the per-tile counts of flagged lookups
// live in the sort's scratch: seg_sort_hyb_tab / seg_sort_bloom / seg_sort_tile_cnt)
struct HybTiles {            // how the bag-major apply tiled the request when it listed the flagged lookups (hyb_rest_kernel reads the lists)
    int bags_per_tile;       // KParams::bags_per_block of the apply
    int tiles_per_table;
};
// seg_sort.hip: the sorted backward's key sort over per-table segments established on the device
constexpr int kSegSortMaxTables = 1024;
#ifndef PM_SEG_TILE
#define PM_SEG_TILE 4096
#endif
constexpr int kSegTile = PM_SEG_TILE;   // elements per radix tile of the key sort (16 per thread); -DPM_SEG_TILE=2048: experiment builds
struct SegDesc {             // one per table, written by the sort's prep kernels (device memory, inside the sort scratch)
    uint32_t in_start;       // first lookup of the (sliced) table in the request's index array
    uint32_t count;          // lookups of the (sliced) table
    uint32_t out_start;      // first position of the table's pairs in the sorted arrays (compact: sum of the counts before)
    uint32_t pooling;        // > 0: every bag of the (sliced) table has exactly this many lookups -- established by reading the offsets
    uint32_t tile_base;      // first 4096-element radix tile of the segment
    uint32_t ntiles;
    uint32_t rbits;          // bits of this table's row ids
    uint32_t pad;
};
struct SegSortRequest {
    const void* indices;
    const void* offsets;
    const int64_t* rows;     // device [T]
    int idx64;
    int T;
    int64_t B, N, bag_begin, bag_count;
    int tshift;              // key = table << tshift | row
    int rbits_max;           // bits_for(max_rows): the number of global passes of mode 0
    bool weighted;           // values = lookup positions (+ bag_of), every table through the key-building kernel
    uint32_t* zero4;         // not NULL: four words the sort's first kernel sets to zero (the apply's work-list control words)
    HybArgs hyb;
    uint32_t* queue_a;       // hybrid: two idle arrays of N words each for the mark kernel's per-slice row queues (the sort's value
    uint32_t* queue_b;       // buffers: nothing else touches them before the bag-major apply); NULL: every slice scans its table
    uint32_t spin_cap;       // look-back polls before a walk gives up (0 = default)
};
size_t seg_sort_scratch_bytes(size_t n_max, int T, bool hybrid_possible = true);   // false (weighted requests): no dup maps / stage
int seg_sort_radix_bits(int mode, int rbits_max);   // 8, or 9 where a 9-bit digit saves a global pass (mode 0)
bool seg_sort_lookback(int mode, int rbits_max, int64_t n);   // mode 0 as one kernel per pass (tiles learn their prefixes from their predecessors in flight)
int seg_sort_passes(int mode, int rbits_max);
bool seg_sort_result_in_b(int mode, int rbits_max);
const SegDesc* seg_sort_desc(const void* scratch, size_t n_max, int T);
const uint32_t* seg_sort_count(const void* scratch, size_t n_max, int T);   // device uint32: pairs in the sorted arrays
// The sort in two parts: part A (the tables' segments and verdicts + the hybrid tables' dup bitmaps: all the bag-major apply
// needs) and part B (everything else).  When the hybrid kernels are launched (rq.hyb.allow) part B runs inside the APPLY call,
// after the bag-major kernel -- which lists the flagged lookups tile by tile as a by-product of its own staging -- and
// hyb_rest_kernel, which finishes the hybrid tables' lists in LDS or compacts them into the a buffers as built pairs (`hybrid_done`
// says that has happened); otherwise seg_sort_pairs = A then B.
template <typename K>
hipError_t seg_sort_part_a(const SegSortRequest& rq, void* scratch, hipStream_t stream);
template <typename K>
hipError_t seg_sort_part_b(const SegSortRequest& rq, int mode, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* bag_of,
                           void* scratch, hipStream_t stream, bool hybrid_done = false);
template <typename K>
hipError_t seg_sort_pairs(const SegSortRequest& rq, int mode, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* bag_of,
                          void* scratch, hipStream_t stream);
const uint32_t* seg_sort_timeouts(const void* scratch, size_t n_max, int T);   // device uint32: look-back walks that gave up
const HybTable* seg_sort_hyb_tab(const void* scratch, size_t n_max, int T);
const uint32_t* seg_sort_bloom(const void* scratch, size_t n_max, int T);
uint32_t* seg_sort_tile_cnt(const void* scratch, size_t n_max, int T);     // [T][B / 4 + 1] at most: flagged lookups per tile of the bag-major apply
size_t seg_sort_tile_cnt_stride(size_t n_max);
uint32_t* seg_sort_rest_stat(void* scratch, size_t n_max, int T);         // two device words: pairs / tables hyb_rest_kernel finished in LDS (zeroed by every sort)
// Round 6: what the bag-major kernel left of a hybrid table -- the flagged lookups it listed tile by tile, ~3 % of a uniform
// request -- is STAGED by hyb_stage_kernel (seg_hybrid.inc): a table with at most kRestCap of them gets its lists copied back to
// back (list = request-position order) as (row, bag) into its own slot of a staging area and its sort segment emptied; a table
// with more gets them compacted into the sort's a buffers as "n built pairs" (what round 5 did for every table).  hyb_rest_kernel
// (embbag_bwd_sorted_kernels.inc) then sorts each staged table inside LDS and applies it run by run; nothing of it reaches the
// key sort, whose launches find no pairs for it.
constexpr int kRestThreads = 1024;
constexpr int kRestItems = 8;                           // pairs per thread in the LDS sort
constexpr int kRestCap = kRestThreads * kRestItems;     // 8192 flagged lookups per table (benchmark shape: ~4 800)
const uint32_t* seg_sort_rest_stage(const void* scratch, size_t n_max, int T);   // [T_h][2][kRestCap]: rows | bags of a staged table
const uint32_t* seg_sort_rest_n(const void* scratch, size_t n_max, int T);       // [T_h]: staged pairs of table t (0: none / not staged)
template <typename K>
hipError_t seg_sort_stage_leftovers(const SegSortRequest& rq, const K* keys_b, const uint32_t* vals_b, K* keys_a, uint32_t* vals_a,
                                    HybTiles tiles, int rest_enable, void* scratch, hipStream_t stream);

// rowquant.hip: row-wise quantisation of fp32 rows (bits 16 / 8 / 4 / 2), dim a multiple of 8
int64_t rows_quantized_row_bytes(int dim, int bits);
hipError_t launch_rows_quantize(const float* src, int64_t n_rows, int dim, int bits, void* dst, hipStream_t stream);
hipError_t launch_rows_dequantize(const void* src, int64_t n_rows, int dim, int bits, float* dst, hipStream_t stream);
void set_backward_tuning(int sort_impl, int order, int xcd, int max_phases);   // -1 = default (environment)
void set_sort_tuning(int mode);                                                // segmented sort: -1 default, 0 / 1 / 2
void set_hybrid_tuning(int enable, uint32_t spin_cap);            // hybrid backward (embbag_bwd_sorted.hip); -1 = default
void set_hybrid_min_tiles(int tiles);                             // ... offered from this many bag-major workgroups on: -1 default (1024)
void set_hybrid_rest(int mode);                                   // ... its left-overs finished in LDS (hyb_rest_kernel): -1 default, 0 / 1
hipError_t sort_status(const KParams& p, int64_t max_rows, int max_dim, const void* workspace, hipStream_t stream, uint32_t out[6]);

// DLRM input redistribution (dlrm_regroup.hip)
hipError_t launch_dlrm_regroup(const int64_t* lengths, const int64_t* indices, int W, int F, int64_t B,
                               int64_t* out_indices, int64_t* out_offsets, int64_t* scratch, hipStream_t stream);

// The lanes of the wave that hold the same NB-bit digit as this lane (gfx9 has no match instruction: one ballot per bit).
// Returns, for a valid lane, how many lower lanes share its digit (`below`) and how many lanes do in all (`total`).
// Written on 32-bit halves: per bit, the sign-extended bit (0 / -1), one compare for the ballot and m &= ~(ballot ^ bit) on
// either half -- as 64-bit selects (`bit ? bal : ~bal`) the compiler spent ~115 vector instructions per element here, and the
// sixteen elements of a thread made this loop half of a pass kernel's time.
template <int NB = 8>
__device__ __forceinline__ void match_digit(uint32_t d, bool valid, uint32_t& below, uint32_t& total) {
    const uint64_t v = __ballot(valid);
    uint32_t xlo = ~static_cast<uint32_t>(v), xhi = ~static_cast<uint32_t>(v >> 32);      // lanes that do NOT match, so far
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int32_t bm = static_cast<int32_t>(d << (31 - b)) >> 31;        // 0 or -1: bit b of the digit
        const uint64_t bal = __ballot(bm != 0);
        // x |= ballot ^ bit: one three-input bit operation per half (truth table 0xF6 = a | (b ^ c))
        xlo = __builtin_amdgcn_bitop3_b32(xlo, static_cast<uint32_t>(bal), static_cast<uint32_t>(bm), 0xF6);
        xhi = __builtin_amdgcn_bitop3_b32(xhi, static_cast<uint32_t>(bal >> 32), static_cast<uint32_t>(bm), 0xF6);
    }
    const uint32_t mlo = ~xlo, mhi = ~xhi;
    below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
    total = static_cast<uint32_t>(__popc(mlo) + __popc(mhi));
}

// the lanes of a wave hand LDS data to each other: LDS operations of one wave execute in order, the compiler must keep them so
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lanes per bag for a given widest row: next power of two >= max_dim / vec, clamped to [8, 64]
inline int group_lanes(int max_dim, int vec) {
    int need = (max_dim + vec - 1) / vec;
    int g = 8;
    while (g < need && g < 64) g <<= 1;
    return g;
}

}  // namespace pm
