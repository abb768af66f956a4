// param_amd/csrc/bwd_sorted_apply.h -- private to the sorted backward: the apply kernels' argument block, the destination
// element types and the launchers.  Included by embbag_bwd_sorted.hip (host logic) and by the three per-dtype translation
// units embbag_bwd_sorted_{f32,bf16,f16}.hip, which hold the kernel instantiations -- 3 dtypes x 2 key widths x 4 lane-group
// sizes x weighted x optimizer x 3 tile sizes of main + fix-up kernels compile in parallel instead of in one 2.5-minute unit.
#pragma once

#include <type_traits>

#include "common.h"

namespace pm {

constexpr int kSortTile = 1024;  // sorted positions per workgroup in step 3 (2048: main +53 us, fix-up -28 us, gpurun s7)

constexpr int kExactRun = 256;       // crossing runs up to this length are re-walked exactly
constexpr int kFixGrid = 2048;      // workgroups of the fix-up kernel: one listed run per lane group and round
constexpr int kMaxTablesLds = 1024;  // per-table metadata staged in LDS up to this many tables

struct ChunkRec {
    uint32_t lead_len;   // positions at the start of the chunk continuing a run begun earlier (0: none); bit 31 (kJoinedBit) on a
                         // tile's first chunk: the whole TILE lies inside one run and that chunk's leading partial is the tile's sum
    uint32_t trail_len;  // positions at the end of the chunk starting a run that continues (0: none)
};

constexpr uint32_t kJoinedBit = 0x80000000u;

struct SortedParams {
    ChunkRec* recs;          // per-chunk piece lengths of boundary-crossing runs (main -> fix-up)
    uint32_t* fix_list;      // chunks holding the head of a run left in pieces, appended by the main kernel
    uint32_t* fix_ctl;       // [0], [1]: list lengths, used in turn; [2], [3]: the generation as main / fix-up kernels read it
    float* partials;         // per chunk: lead / trail fp32 partial sums, 2 * max_dim floats
    int32_t T;
    const void* keys;        // sorted keys (uint32 / uint64)
    const uint32_t* vals;    // sorted values: bag within table (unweighted) or lookup position j (weighted)
    const uint32_t* bag_of;  // weighted only: bag within table of lookup position j
    void* const* dst;        // destination tables
    const int32_t* dims;
    const int64_t* out_offsets;
    const float* grad;
    const float* psw;
    int64_t out_stride;
    int32_t gblk_shift;      // blocked gradient layout (common.h: grad_bag_offset); extra == 0: none
    int64_t gblk_extra;
    int64_t n;               // number of sorted pairs
    int32_t rbits;           // key = (t << rbits) | row ; keys with bit (tbits+rbits) set are padding
    int32_t kbits;           // tbits + rbits
    int32_t max_dim;
    int32_t nt_rows;         // 1: streaming (non-temporal) destination-row loads/stores
    float alpha;
    float* const* mom;       // row-wise Adagrad: device array [T] of per-row fp32 state (else NULL)
    float lr;
    float eps;
    float wd;                // weight decay (row-wise Adagrad)
    int32_t wd_mode;         // PM_WD_NONE / PM_WD_L2 / PM_WD_DECOUPLE
    int32_t sr;              // 1: stochastic rounding of the updated row (16-bit tables)
    uint64_t sr_seed;
    int32_t exact_run;       // crossing runs up to this length are re-walked exactly in the fix-up
    int32_t tshift;          // table id = key >> tshift (rbits + phase bits)
    int32_t seg_tiles;       // > 0: the sorted array is T * H equal segments of this many tiles (fixed pooling)
    int32_t H;               // bag phases (1 or 2): segments are (table, phase), one apply launch per phase
    int32_t phase;           // phase this launch applies
    int32_t tile;            // sorted positions per workgroup of the apply kernels (kSortTile, or smaller for small requests)
    int32_t xcd;             // 1: XCD-affine block -> tile mapping (needs seg_tiles); 2: XCD-contiguous (any request)
    const uint32_t* d_n;     // not NULL: the number of sorted pairs lives on the device (<= n), written by the segmented sort
    int32_t unique_wgs_per_cu;   // bag-major apply: > 0 = a grid of this many workgroups per CU that loop over the tiles (0: one per tile)
    int32_t join_tiles;      // 1: a tile that lies wholly inside one run hands the fix-up ONE partial sum instead of one per chunk (set by
                             // the apply's launcher: it needs every lane group to make the same number of column passes)
};


// the apply of one destination dtype (defined in embbag_bwd_sorted_<dtype>.hip)
hipError_t bwd_sorted_launch_f32(const SortedParams& sp, int key_bytes, int max_dim, hipStream_t stream);
hipError_t bwd_sorted_launch_bf16(const SortedParams& sp, int key_bytes, int max_dim, hipStream_t stream);
hipError_t bwd_sorted_launch_f16(const SortedParams& sp, int key_bytes, int max_dim, hipStream_t stream);
// the hybrid backward's bag-major apply of rows looked up once (bwd_unique_kernel); kp = the request with the forward's tiling
struct UniqueArgs {
    const HybTable* hyb_tab;
    const uint32_t* bloom;
    void* emit_keys;             // the sort's b buffers: the flagged lookups of tile (t, i) go to the tile's own request positions
    uint32_t* emit_vals;
    int key_bytes;
    uint32_t* tile_cnt;          // [table][tile_cnt_stride]
    size_t tile_cnt_stride;
    int bloom_wbits;             // log2 of the words of a table's dup map (hyb_slices x 16 384 words)
};
hipError_t bwd_unique_launch_f32(const SortedParams& sp, const KParams& kp, const UniqueArgs& ua, int max_dim, hipStream_t stream);
hipError_t bwd_unique_launch_bf16(const SortedParams& sp, const KParams& kp, const UniqueArgs& ua, int max_dim, hipStream_t stream);
hipError_t bwd_unique_launch_f16(const SortedParams& sp, const KParams& kp, const UniqueArgs& ua, int max_dim, hipStream_t stream);

// Round 6: the hybrid tables' left-overs that hyb_stage_kernel staged (common.h) -- sorted in LDS by each of `parts` workgroups of a
// table (1024 threads, stable 8-bit rounds on the row bits) and applied run by run in place, every workgroup a slice of the sorted
// positions.  Before: compact 8 + prep 2 5 + histogram 6.5 + scan 5 + 3 passes 28 + sorted apply 65 + fix-up 5 us for the 226 K pairs
// of the uniform benchmark request.
struct RestArgs {
    const HybTable* hyb_tab;
    int T_h;                     // tables that can be hybrid (min(T, kHybMaxTables))
    const uint32_t* stage;       // [T_h][2][kRestCap]
    const uint32_t* rest_n;      // [T_h]
    const uint32_t* rbits;       // &SegDesc[0].rbits, stride sizeof(SegDesc) / 4 words: bits of table t's row ids
    int rbits_stride;
    int parts;                   // workgroups per table
};
int rest_parts(int T_h, int tiles_per_table);
hipError_t bwd_rest_launch_f32(const SortedParams& sp, const KParams& kp, const RestArgs& ra, int max_dim, hipStream_t stream);
hipError_t bwd_rest_launch_bf16(const SortedParams& sp, const KParams& kp, const RestArgs& ra, int max_dim, hipStream_t stream);
hipError_t bwd_rest_launch_f16(const SortedParams& sp, const KParams& kp, const RestArgs& ra, int max_dim, hipStream_t stream);

}  // namespace pm
