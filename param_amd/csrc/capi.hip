// param_amd/csrc/capi.hip -- extern "C" entry points of libparam_amd.so (include/param_amd.h).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "common.h"

namespace {

thread_local std::string g_last_error;

// Row loads a lane group keeps in flight in the forward.  Round 3 found that rounds 1-2 had measured this knob with ONE load
// in flight whatever its value (a wait at the merge of the staged / unstaged index paths, embbag_fwd.hip); with the batch
// real: 1 -> 19.3 G lookups/s Zipf / 0.727 of the HBM peak uniform, 2 -> 22.2 / 0.734, 3 -> 21.5 / 0.712, 4 -> 21.7 / 0.732,
// 6 -> 21.3 / 0.712, 8 -> 21.4 / 0.732 (48 x 10 M x 128 fp32, profiles/r03_fwd_unroll_sweep.txt); Criteo 14.1 / 15.6 / 15.1 at 1 / 2 / 4.
constexpr int kDefaultUnroll = 2;
std::atomic<int> g_unroll{0};
std::atomic<int> g_bags_per_block{0};
std::atomic<int> g_xcd_affine{-1};
std::atomic<int> g_nt_loads{-1};
std::atomic<int> g_stage_out{-1};
std::atomic<int> g_flat_target{-1}; // pm_set_forward_tuning: lookups per flat-walk tile (-1: PARAM_AMD_FLAT_TARGET, default 256)
std::atomic<int> g_flat_grid{-1};   // pm_set_forward_tuning: launch shape of the flat-walk forward (-1: PARAM_AMD_FLAT_COMPACT, default 1)
// destination-row cache policy of the sorted backward when pm_set_tuning leaves nt_loads at its default: plain loads,
// agent-scope (sc1) stores.  The store writes through and drops the row's lines from the XCD's L2, so a row occupies L2 only
// between its load and its store and the gradient rows -- re-read once per lookup of their bag -- keep the capacity
// (round 3, 48 x 10 M x 128 fp32, uniform indices: apply 1.58 -> 1.49 ms; Zipf 0.953 -> 0.943 ms).
constexpr int kDefaultRowPolicy = 3;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t rc, const char* what) {
    return fail(PM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(rc));
}

// The forward's environment switches (kernel sweeps; every one has a pm_set_* or a default that the product uses): read ONCE
// per process, on the first forward -- not on the launch path (getenv is not thread-safe against setenv, and a value that
// changes between two calls of one request's sequence is a hazard, not a feature).
struct FwdEnv {
    int stage;        // PARAM_AMD_FWD_STAGE=0: no LDS-staged output burst
    int flat;         // PARAM_AMD_FWD_FLAT=0: no flat-walk kernel; 2: also for pooling factors up to 4
    int flat_maxl;    // PARAM_AMD_FLAT_MAXL
    int flat_target;  // PARAM_AMD_FLAT_TARGET (lookups per flat-walk tile)
    int flat_bags;    // PARAM_AMD_FLAT_BAGS (bags per flat-walk tile at most)
    int tile_major;   // PARAM_AMD_FWD_TILE_MAJOR=1
    int flat_compact; // PARAM_AMD_FLAT_COMPACT=0: the flat-walk kernel's old grid (T x smallest-tile count); N > 1: exactly N workgroups
};
const FwdEnv& fwd_env() {
    static const FwdEnv e = [] {
        auto num = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
        FwdEnv r;
        const char* st = getenv("PARAM_AMD_FWD_STAGE");
        r.stage = (st && st[0] == '0') ? 0 : 1;
        r.flat = num("PARAM_AMD_FWD_FLAT", 1);
        r.flat_maxl = num("PARAM_AMD_FLAT_MAXL", 0);
        r.flat_target = num("PARAM_AMD_FLAT_TARGET", 256);
        r.flat_bags = num("PARAM_AMD_FLAT_BAGS", 256);
        r.tile_major = num("PARAM_AMD_FWD_TILE_MAJOR", -1);
        r.flat_compact = num("PARAM_AMD_FLAT_COMPACT", 1);
        return r;
    }();
    return e;
}

int flat_target_knob(const FwdEnv& env) { const int k = g_flat_target.load(); return k > 0 ? k : env.flat_target; }

bool dtype_is_weight(int d) { return d == PM_F32 || d == PM_BF16 || d == PM_F16; }

// Validate the host-visible part of a request and derive the launch geometry.
// forward = the call is pm_embbag_fwd / pm_embbag_fwd_quantized: only those get the staged-output geometry (smaller tiles,
// index tile sized for the request) and work tiles; every other entry point keeps the plain bag-count tiling.
int make_params(const pm_embbag_batch* op, int elem_dtype, pm::KParams& p, bool forward = false) {
    if (!op) return fail(PM_ERR_INVALID, "op is NULL");
    if (op->num_tables < 1) return fail(PM_ERR_INVALID, "num_tables must be >= 1");
    if (!dtype_is_weight(elem_dtype)) return fail(PM_ERR_INVALID, "weight/dst dtype must be PM_F32, PM_BF16 or PM_F16");
    if (op->index_dtype != PM_I64 && op->index_dtype != PM_I32)
        return fail(PM_ERR_INVALID, "index_dtype must be PM_I64 or PM_I32");
    if (op->batch < 0 || op->num_indices < 0) return fail(PM_ERR_INVALID, "negative batch / num_indices");
    if (op->bag_begin < 0 || op->bag_count < 0 || op->bag_begin + op->bag_count > op->batch)
        return fail(PM_ERR_INVALID, "bag_begin/bag_count outside [0, batch]");
    if (!op->tables || !op->rows || !op->dims || !op->out_offsets)
        return fail(PM_ERR_INVALID, "tables/rows/dims/out_offsets must be device pointers");
    if (op->bag_count > 0 && !op->offsets) return fail(PM_ERR_INVALID, "offsets is NULL");
    if (op->num_indices > 0 && !op->indices) return fail(PM_ERR_INVALID, "indices is NULL");
    const int vec = (elem_dtype == PM_F32) ? 4 : 8;
    if (op->max_dim < 1 || op->max_dim % vec != 0)
        return fail(PM_ERR_UNSUPPORTED, "every dims[t] (and max_dim) must be a multiple of " +
                                            std::to_string(vec) + " for this element type");
    if (op->out_stride % 4 != 0) return fail(PM_ERR_UNSUPPORTED, "out_stride must be a multiple of 4 elements");
    if (op->min_dim < 0 || op->min_dim > op->max_dim || (op->min_dim > 0 && op->min_dim % vec != 0))
        return fail(PM_ERR_INVALID, "min_dim must be 0 (not given) or a multiple of the vector width in [1, max_dim]");
    if (op->fixed_pooling < 0 ||
        (op->fixed_pooling > 0 && op->fixed_pooling * op->batch * static_cast<int64_t>(op->num_tables) != op->num_indices))
        return fail(PM_ERR_INVALID, "fixed_pooling must be 0 or num_indices / (num_tables * batch)");

    const int G = pm::group_lanes(op->max_dim, vec);
    const int NG = pm::kBlock / G;
    int bpb = g_bags_per_block.load();
    if (bpb <= 0) {
        // 4 bags per lane group amortise the LDS staging; small requests shrink the tile until the
        // grid fills the chip (256 CUs x 8 resident workgroups) -- a 16 K-bag single-table lookup
        // ran 2 workgroups per CU at 49 % of the roofline with the fixed tile (gpurun r1e)
        bpb = 4 * NG;
        // Short average bags go with one bag per lane group: requests whose tables have very different bag
        // sizes (Criteo multi-hot 1 .. 100, average 8.2) otherwise leave a few tables' workgroups 100x longer
        // than the rest -- measured 7.2 (32 bags/tile) vs 14.3 G lookups/s (8); with uniform L = 20 the
        // 32-bag tile is 3.5 % faster under Zipf and 2 % slower under uniform indices (sweeps r1f/r1g).
        {
            const int64_t tb = static_cast<int64_t>(op->num_tables) * op->batch;
            if (tb > 0 && op->num_indices < 12 * tb) bpb = NG;
        }
        const int64_t want_blocks = 256 * 8;
        while (bpb > NG && static_cast<int64_t>(op->num_tables) * ((op->bag_count + bpb - 1) / bpb) < want_blocks) bpb /= 2;
    }
    if (bpb < NG) bpb = NG;
    if (bpb > 1024) bpb = 1024;

    const int64_t tiles = (op->bag_count + bpb - 1) / bpb;
    if (tiles * op->num_tables > 0x7fffffffLL) return fail(PM_ERR_UNSUPPORTED, "grid too large");

    const int64_t total_bags = static_cast<int64_t>(op->num_tables) * op->batch;
    const int64_t avg_l = total_bags > 0 ? (op->num_indices + total_bags - 1) / total_bags : 0;
    int64_t cap = 2 * bpb * avg_l;
    cap = (cap + 255) / 256 * 256;
    // tables of one request can have very different bag sizes (Criteo multi-hot 1 .. 100 at an average of 8):
    // size the tile for 4x the average, at least 2048 entries (4096 unweighted: 16 KiB of LDS per workgroup)
    if (cap < (op->per_sample_weights ? 2048 : 4096)) cap = op->per_sample_weights ? 2048 : 4096;
    if (cap > 4096) cap = 4096;

    p.tables = op->tables;
    p.rows = op->rows;
    p.dims = op->dims;
    p.out_offsets = op->out_offsets;
    p.indices = op->indices;
    p.offsets = op->offsets;
    p.psw = op->per_sample_weights;
    p.io = nullptr;
    p.out_stride = op->out_stride;
    p.B = op->batch;
    p.N = op->num_indices;
    p.bag_begin = op->bag_begin;
    p.bag_count = op->bag_count;
    p.T = op->num_tables;
    p.tiles_per_table = static_cast<int32_t>(tiles);
    p.bags_per_block = bpb;
    p.idx_cap = static_cast<int32_t>(cap);
    p.idx64 = op->index_dtype == PM_I64 ? 1 : 0;
    const int xa = g_xcd_affine.load();
    // T % 8 == 0: table t on XCD t % 8.  Any other table count (26 tables: BASELINE configs[3]): contiguous eighths of the
    // table-major tile order (common.h) -- 26 x 10 M x 128, [B, sum D] output: 0.692 -> 0.700 of the HBM peak under uniform indices,
    // 19.6 -> 21.15 G lookups/s under Zipf (tools/r4_fwd_xcd.py) -- but only where every table has the same lookups per tile: with
    // the Criteo tables' pooling factors of 1 .. 100 the XCD that gets the 100-hot table does a third of the launch (0.70 -> 0.27)
    {
        const int64_t tb = static_cast<int64_t>(op->num_tables) * op->batch;
        const bool even_req = tb > 0 && op->num_indices % tb == 0;
        p.xcd_affine = xa == 0 ? 0 : (op->num_tables % pm::kXcds == 0 ? 1 : ((op->num_tables > 1 && even_req) ? 3 : 0));
        // blocked requests (table_group = W consecutive request tables read ONE weight table): contiguous eighths of the
        // table-major tile order keep a weight table's blocks on one XCD; t % 8 would spread them over all eight
        if (xa != 0 && op->table_group > 1 && op->num_tables > 1 && even_req) p.xcd_affine = 3;
    }
    const int nt = g_nt_loads.load();
    p.nt_loads = nt > 0 ? nt : 0;   // forward: any non-zero = non-temporal row loads; sorted backward: 1 nt, 2 system scope
    // lookups that do not divide evenly over the bags: certainly ragged (fixed-size requests -- every benchmark shape --
    // always divide; a ragged request that happens to divide merely runs the unordered kernel)
    // ... and only where the order can matter: with one bag per lane group (short-bag tiles) nothing is pulled
    p.ordered = (total_bags > 0 && op->num_indices % total_bags != 0 && bpb > NG) ? 1 : 0;
    p.alpha = 1.0f;
    if (op->grad_block_shift < 0 || op->grad_block_shift > 31 || op->grad_block_extra < 0 || op->table_group < 0)
        return fail(PM_ERR_INVALID, "grad_block_shift must be in [0, 31], grad_block_extra and table_group >= 0");
    // a blocked gradient needs blocks of at least two bags that tile the batch (shift 0 with a non-zero extra used to read
    // the un-blocked layout without a word, and so did a batch that is not a whole number of blocks)
    if (op->grad_block_shift == 0 && op->grad_block_extra != 0)
        return fail(PM_ERR_INVALID, "grad_block_extra needs grad_block_shift >= 1 (a block of one bag is the un-blocked layout)");
    if (op->grad_block_shift > 0 && (op->batch & ((static_cast<int64_t>(1) << op->grad_block_shift) - 1)) != 0)
        return fail(PM_ERR_INVALID, "batch must be a multiple of 2^grad_block_shift bags");
    p.gblk_shift = op->grad_block_shift > 0 ? op->grad_block_shift : 31;
    p.gblk_extra = op->grad_block_shift > 0 ? op->grad_block_extra : 0;
    // LDS-staged output (forward): the tile's pooled rows leave in one burst at the end of the tile (embbag_fwd.hip).
    // Measured at benchmark size: uniform indices 0.69 -> 0.72-0.74 of the HBM peak in both output layouts, Zipf within
    // +-2 % -- provided the workgroup's LDS stays small (at 6 workgroups per CU the latency-bound Zipf launch lost 12 %):
    // so only for requests whose lookups divide evenly over the bags (fixed pooling: every benchmark shape), whose index
    // tile can then be sized for what a tile really holds instead of the 4096-entry default, and only where the staging
    // buffer fits 16 KB (the tile is halved until it does).  pm_set_forward_tuning(0) / PARAM_AMD_FWD_STAGE=0 turn it off.
    p.out_bits = 0;
    p.stage_out = 0;
    p.stage_bags = 0;
    p.flat_bags = 0;
    p.flat_target = 0;
    p.flat_compact = 0;
    if (forward) {
        const FwdEnv& env = fwd_env();
        int want = g_stage_out.load();
        if (want < 0) want = env.stage;
        const bool even = total_bags > 0 && op->num_indices % total_bags == 0;
        if (want && even && !p.ordered && g_bags_per_block.load() <= 0) {
            while (bpb > NG && static_cast<int64_t>(bpb) * op->max_dim * 4 > 16384) bpb /= 2;
            if (static_cast<int64_t>(bpb) * op->max_dim * 4 <= 16384) {
                const int64_t tiles2 = (op->bag_count + bpb - 1) / bpb;
                if (tiles2 * op->num_tables <= 0x7fffffffLL) {
                    p.bags_per_block = bpb;
                    p.tiles_per_table = static_cast<int32_t>(tiles2);
                    p.stage_out = op->max_dim;
                    int64_t need = (2 * static_cast<int64_t>(bpb) * avg_l + 255) / 256 * 256;
                    if (need < 512) need = 512;
                    if (need < p.idx_cap) p.idx_cap = static_cast<int32_t>(need);
                }
            }
        } else if (want && even && !p.ordered && static_cast<int64_t>(bpb) * op->max_dim * 4 <= 16384) {
            p.stage_out = op->max_dim;   // explicit bags_per_block (sweeps): stage if it fits, index tile sized the same way
            int64_t need = (2 * static_cast<int64_t>(bpb) * avg_l + 255) / 256 * 256;
            if (need < 512) need = 512;
            if (need < p.idx_cap) p.idx_cap = static_cast<int32_t>(need);
        } else if (want && !even && !p.ordered && g_bags_per_block.load() <= 0 &&
                   static_cast<int64_t>(bpb) * op->max_dim * 4 <= 4096) {
            // short-bag tiles of requests with per-table pooling (Criteo multi-hot: one bag per lane group, 8 bags per
            // tile): the staging buffer is 4 KB, the full-size index tile stays -- 20.6 KB per workgroup
            p.stage_out = op->max_dim;
        }
        p.stage_bags = p.bags_per_block;
        // ... and those requests run the flat-walk kernel (embbag_fwd.hip): tiles sized per table on the device (~256 lookups, up
        // to 32 bags for the pooling-1 tables), row loads in flight across bag borders.  tiles_per_table stays the count of the
        // smallest tile (one bag per lane group): workgroups past their table's tile count leave.  Criteo tables, visit
        // r3_criteo_flat (Zipf G lookups/s / uniform fraction; run-to-run +-1.5 %): 8-bag tiles 15.7-15.9 / 0.69-0.71; flat walk
        // target 512 cap 64: 16.4-16.5 / 0.705-0.716; **256 / 32: 16.6 / 0.719**; 128: 16.1 / 0.695; 1024 / 128: 14.1 / 0.64.
        // PARAM_AMD_FWD_FLAT=0 turns it off (PARAM_AMD_FLAT_TARGET / _BAGS: sweeps).
        // (Round 6: that cap of 32 bags was chosen under round 3's grid, where a larger cap also meant more surplus workgroups.  With
        // the compact launch a one-hot table's 32-bag tile is three round trips of set-up around 32 row loads; one process per setting,
        // three each, us uniform / Zipf: cap 32: 192-193 / 118; 64: 182 / 114; 128: 186 / 112; **256: 181 / 111.6**
        // -- profiles/r06_flat_bags_cap_sweep_all128.jsonl.  The cap is 256 bags on every flat-walk path now.)
        const int flat_env = env.flat;
        const bool flat_on = flat_env != 0;
        // ... and so do fixed-pooling requests of one or two lookups per bag (one-hot tables): bag by bag a lane group has one or
        // two row loads in flight.  48 x 10 M x 128 fp32, batch 65536 (tools/r3_shortbags.sh; Zipf G lookups/s / uniform fraction):
        // pooling 1: 4.00 / 0.529 -> 5.05 / 0.624; pooling 2: 7.78 / 0.687 -> 8.75 / 0.686; pooling 4: 13.5 / 0.723 -> 13.4 / 0.679
        // (not taken; PARAM_AMD_FWD_FLAT=2 extends the rule to 4 for that measurement).
        const int flat_maxl = env.flat_maxl;
        if (flat_on && even && avg_l <= (flat_maxl > 0 ? flat_maxl : flat_env == 2 ? 4 : 2) && !p.ordered && g_bags_per_block.load() <= 0 &&
            p.stage_out > 0) {
            const int64_t tiles_ng = (op->bag_count + NG - 1) / NG;
            if (tiles_ng * op->num_tables <= 0x7fffffffLL) {
                const int tgt2 = flat_target_knob(env);
                p.flat_bags = env.flat_bags < 32 ? 32 : (env.flat_bags > 1024 ? 1024 : env.flat_bags / NG * NG);
                p.flat_target = tgt2;
                p.tiles_per_table = static_cast<int32_t>(tiles_ng);
                p.stage_bags = NG;
                p.bags_per_block = p.flat_bags;
                if (p.idx_cap < 1024) p.idx_cap = 1024;
            }
        }
        if (flat_on && !even && !p.ordered && g_bags_per_block.load() <= 0 && bpb == NG && p.stage_out > 0) {
            const int cap_env = env.flat_bags;
            const int tgt_env = flat_target_knob(env);
            p.flat_bags = cap_env < NG ? NG : (cap_env > 1024 ? 1024 : cap_env / NG * NG);
            p.flat_target = tgt_env < 1 ? 1 : tgt_env;
            p.bags_per_block = p.flat_bags;   // sizes the LDS offsets array; stage_bags (the burst buffer) stays at NG rows
        }
        // ABI v7, mixed embedding dims (min_dim given and its lane group narrower than max_dim's): the flat-walk kernel sizes the lane
        // group PER TABLE on the device (embbag_fwd.hip), whatever the bags look like -- tiles are ~flat_target lookups of any pooling
        // factor.  flat_bags = the bags the narrowest table's sub-groups pool at once (at least the usual 32): the LDS offsets array.
        // Ragged requests keep the ordered kernel (longest bag first matters more there than idle lanes).
        if (flat_on && op->min_dim > 0 && op->min_dim < op->max_dim && !p.ordered && g_bags_per_block.load() <= 0) {
            int need = (op->min_dim + vec - 1) / vec, g_min = 4;          // sub-groups of 4 lanes at least: 64-byte pieces
            while (g_min < need) g_min <<= 1;
            if (g_min < G) {
                const int ng_max = pm::kBlock / g_min;
                const int64_t tiles_ng = (op->bag_count + NG - 1) / NG;   // the grid of the smallest tile: the widest table's NG bags
                if (tiles_ng * op->num_tables <= 0x7fffffffLL) {
                    if (p.flat_bags < ng_max) p.flat_bags = ng_max;
                    // (256 bags at most per tile: a one-hot narrow table then reaches the ~flat_target lookups per tile the wide tables
                    // have -- its 64-bag tiles were three round trips of set-up around 4 KB of rows.  Criteo tables with mixed dims, one
                    // process per setting, two each: cap 64: 143.3-143.9 us, 128: 142.3-142.9, 256: 141.5-142.1; the narrow tables alone
                    // 24.3 -> 21.6 us: profiles/r06_flat_bags_cap_sweep.jsonl)
                    if (p.flat_bags < 256) p.flat_bags = 256;
                    if (p.flat_bags < 32) p.flat_bags = 32;
                    if (p.flat_target <= 0) p.flat_target = flat_target_knob(env) < 1 ? 1 : flat_target_knob(env);
                    p.tiles_per_table = static_cast<int32_t>(tiles_ng);
                    p.bags_per_block = p.flat_bags;
                    p.stage_out = op->max_dim;
                    p.stage_bags = NG;                                    // burst buffer: NG rows of max_dim floats (4 KB at D = 128 fp32)
                    // index tile: twice what a tile holds on average -- ~flat_target lookups, or the narrowest table's ng_max bags
                    int64_t cap2 = 2 * static_cast<int64_t>(ng_max) * avg_l;
                    if (cap2 < 2 * static_cast<int64_t>(p.flat_target)) cap2 = 2 * static_cast<int64_t>(p.flat_target);
                    cap2 = (cap2 + 255) / 256 * 256;
                    p.idx_cap = static_cast<int32_t>(cap2 < 1024 ? 1024 : (cap2 > 4096 ? 4096 : cap2));
                    if (p.xcd_affine == 3) p.xcd_affine = 0;             // eighths of the tile order assume equal tiles per table
                }
            }
        }
        // flat-walk launches are one resident set of workgroups walking the tile order (embbag_fwd.hip, round 6): 1 = the launcher asks
        // the device how many that is
        {
            int fg = g_flat_grid.load();
            if (fg < 0) fg = env.flat_compact;
            if (p.flat_bags > 0 && op->num_tables <= 1024 && fg > 0) {
                // the tiles the request will have, from its sizes (the tables' own pooling factors are on the device): ~flat_target
                // lookups each, but at most flat_bags bags; a quarter more, because a table's tile is its average bag count rounded DOWN
                // to whole lane-group rounds
                const int64_t n_slice = op->batch > 0 ? (op->num_indices * op->bag_count + op->batch - 1) / op->batch : 0;
                const int64_t by_lookups = (n_slice + p.flat_target - 1) / p.flat_target;
                const int64_t by_bags = static_cast<int64_t>(op->num_tables) * ((op->bag_count + p.flat_bags - 1) / p.flat_bags);
                int64_t est = (by_lookups > by_bags ? by_lookups : by_bags) * 5 / 4 + op->num_tables;
                if (est < 1024) est = 1024;
                if (est > (1 << 20)) est = 1 << 20;
                p.flat_compact = fg > 1 ? fg : static_cast<int32_t>(est);
            }
        }
        // Two tilings built and measured in round 3 for requests whose tables have very different pooling factors (Criteo
        // multi-hot 1 .. 100) -- both slower than the 8-bag tiles in table-major order, which stay:
        //  * WORK tiles (~640 lookups per tile whatever the pooling factor, tile boundaries derived on the device from the
        //    tables' lookup counts): 186 us against 120 us under Zipf, 249 against 185 us under uniform indices (caps of
        //    32 .. 1024 bags, targets of 320 .. 1280: all slower).  A pooling-1 bag is ONE row load; a lane group walking 16 ..
        //    128 of them in turn has one load in flight where 1024 eight-bag workgroups have them all in flight at once -- the
        //    small tiles ARE the memory-level parallelism.  Removed again (profiles/r03_criteo_fwd_work_tiles.txt).
        //  * tile-major block order (t = b % T, so every table advances at the same rate and the heavy table's long workgroups
        //    start throughout the launch): 165 against 120 us under Zipf, 220 against 186 us uniform -- a table's tiles then run
        //    on all eight XCDs at once and on every CU next to other tables' rows.  Kept behind PARAM_AMD_FWD_TILE_MAJOR=1.
        const int tm_env = env.tile_major;
        const bool uneven = total_bags > 0 && op->num_indices % total_bags != 0;
        if (p.xcd_affine != 1 && op->num_tables > 1 && tm_env > 0 && uneven) p.xcd_affine = 2;
    }
    return PM_OK;
}

// Row loads in flight per lane group when pm_set_tuning leaves the choice to the library.  Two is the measured optimum of a launch
// that fills the chip (eight workgroups per CU: occupancy provides the memory-level parallelism).  A SMALL request -- the reference
// driver's batch 512 .. 4096 rows of one 14 M x 128 table, train/compute/pt/dataset.py:56-82 -- has one or two workgroups per CU,
// each lane group a chain of pooling / 2 dependent round trips: there a deeper batch IS the parallelism.  Additions stay in index
// order: same bits for every value.
int small_request_unroll(const pm_embbag_batch* op, const pm::KParams& p) {
    if (p.flat_bags > 0 || p.ordered) return kDefaultUnroll;
    const int64_t wgs = static_cast<int64_t>(p.T) * p.tiles_per_table;
    const int64_t bags = static_cast<int64_t>(op->num_tables) * op->batch;
    const int64_t avg_l = bags > 0 ? op->num_indices / bags : 0;
    // kernel us under rocprofv3 --kernel-trace (profiles/r06_small_batch_kernel_us.txt; batch 2 / 4 / 8): 64 workgroups 6.6 / 5.6 / 5.5,
    // 128: 6.9 / 6.0 / 5.6, 256: 9.6 / 7.7 / 6.6, 512: 13.0 / 13.1 / 13.1, 1024: 22.4 / 22.6 / 23.1 -- it pays up to one workgroup per CU
    if (wgs > 384) return kDefaultUnroll;
    return avg_l >= 8 ? 8 : (avg_l >= 4 ? 4 : kDefaultUnroll);
}

}  // namespace

extern "C" {

int pm_abi_version(void) { return PM_ABI_VERSION; }

const char* pm_build_info(void) {
    return "libparam_amd gfx950 (CDNA4, wave64) hip " __VERSION__;
}

const char* pm_last_error(void) { return g_last_error.c_str(); }

int pm_set_tuning(int32_t unroll, int32_t bags_per_block, int32_t xcd_affine, int32_t nt_loads) {
    if (unroll != 0 && unroll != 1 && unroll != 2 && unroll != 3 && unroll != 4 && unroll != 6 && unroll != 8)
        return fail(PM_ERR_INVALID, "unroll must be 0, 1, 2, 3, 4, 6 or 8");
    if (bags_per_block < 0) return fail(PM_ERR_INVALID, "bags_per_block must be >= 0");
    g_unroll.store(unroll);
    g_bags_per_block.store(bags_per_block);
    g_xcd_affine.store(xcd_affine);
    g_nt_loads.store(nt_loads);
    return PM_OK;
}

int pm_set_forward_tuning(int32_t stage_out, int32_t flat_grid, int32_t flat_target) {
    if (stage_out < -1 || stage_out > 1) return fail(PM_ERR_INVALID, "stage_out must be -1, 0 or 1");
    if (flat_grid < -1 || flat_grid > (1 << 20)) return fail(PM_ERR_INVALID, "flat_grid must be -1 (default), 0, 1 or a workgroup count up to 2^20");
    if (flat_target < -1 || flat_target == 0 || flat_target > 4096) return fail(PM_ERR_INVALID, "flat_target must be -1 (default) or 1 .. 4096");
    g_stage_out.store(stage_out);
    g_flat_grid.store(flat_grid);
    g_flat_target.store(flat_target);
    return PM_OK;
}

int pm_set_backward_tuning(int32_t sort_impl, int32_t order, int32_t xcd_affine, int32_t max_phases) {
    if (sort_impl < -1 || sort_impl > 2 || order < -1 || order > 1 || xcd_affine < -1 || xcd_affine > 1)
        return fail(PM_ERR_INVALID, "sort_impl must be -1 .. 2, order / xcd_affine -1, 0 or 1");
#ifndef PM_ALTERNATES
    if (sort_impl > 0)
        return fail(PM_ERR_UNSUPPORTED, "sort_impl 1 (rocPRIM) and 2 (round 2's LSD sort) exist in the alternates build only "
                                        "(libparam_amd_alt.so: make -C param_amd/csrc alt)");
#endif
    if (max_phases != -1 && max_phases != 1 && max_phases != 2) return fail(PM_ERR_INVALID, "max_phases must be -1, 1 or 2");
    pm::set_backward_tuning(sort_impl, order, xcd_affine, max_phases);
    return PM_OK;
}

int pm_set_sort_tuning(int32_t mode) {
    if (mode < -1 || mode > 3) return fail(PM_ERR_INVALID, "mode must be -1, 0, 1, 2 or 3");
    pm::set_sort_tuning(mode);
    return PM_OK;
}

int pm_set_hybrid_tuning(int32_t enable, int64_t lookback_spin_cap) {
    if (enable < -1 || enable > 2) return fail(PM_ERR_INVALID, "enable must be -1 .. 2");
    if (lookback_spin_cap < 0 || lookback_spin_cap > 0xffffffffLL) return fail(PM_ERR_INVALID, "lookback_spin_cap must be in [0, 2^32)");
    pm::set_hybrid_tuning(enable, static_cast<uint32_t>(lookback_spin_cap));
    return PM_OK;
}

int pm_set_hybrid_rest(int32_t mode) {
    if (mode < -1 || mode > 1) return fail(PM_ERR_INVALID, "mode must be -1 (default), 0 or 1");
    pm::set_hybrid_rest(mode);
    return PM_OK;
}

int pm_set_hybrid_min_tiles(int32_t tiles) {
    if (tiles < -1) return fail(PM_ERR_INVALID, "tiles must be -1 (default) or >= 0");
    pm::set_hybrid_min_tiles(tiles);
    return PM_OK;
}

int pm_embbag_sort_status(const pm_embbag_batch* op, int64_t max_rows, const void* workspace, pm_sort_status* out, pm_stream_t stream) {
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p);
    if (rc != PM_OK) return rc;
    if (max_rows < 1 || max_rows > (1LL << 31)) return fail(PM_ERR_INVALID, "max_rows must be in [1, 2^31]");
    if (!workspace || !out) return fail(PM_ERR_INVALID, "NULL argument");
    uint32_t v[6] = {0, 0, 0, 0, 0, 0};
    const hipError_t h = pm::sort_status(p, max_rows, op->max_dim, workspace, static_cast<hipStream_t>(stream), v);
    if (h == hipErrorInvalidValue) return fail(PM_ERR_INVALID, "no sort has been recorded for this workspace");
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_sort_status");
    out->lookback_fallbacks = v[0];
    out->pairs_sorted = v[1];
    out->hybrid_tables = v[2];
    out->hybrid_launched = v[3];
    out->lds_pairs = v[4];
    out->lds_tables = v[5];
    return PM_OK;
}

#ifdef PM_ALTERNATES
int64_t pm_radix_sort_scratch_bytes(int64_t n_max) {
    if (n_max < 0 || n_max > 0xffffffffLL) return fail(PM_ERR_INVALID, "n_max must be in [0, 2^32)");
    return static_cast<int64_t>(pm::rs_scratch_bytes(static_cast<size_t>(n_max)));
}

int pm_radix_sort_pairs(void* keys_a, void* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t n_max,
                        const uint32_t* d_count, int32_t key_bytes, int32_t begin_bit, int32_t end_bit, int64_t segment_len,
                        void* scratch, int64_t scratch_bytes, int32_t* result_in_b, pm_stream_t stream) {
    if (n_max < 0 || n_max > 0xffffffffLL) return fail(PM_ERR_INVALID, "n_max must be in [0, 2^32)");
    if (key_bytes != 4 && key_bytes != 8) return fail(PM_ERR_INVALID, "key_bytes must be 4 or 8");
    if (begin_bit < 0 || end_bit < begin_bit || end_bit > key_bytes * 8) return fail(PM_ERR_INVALID, "bad bit range");
    if (!result_in_b) return fail(PM_ERR_INVALID, "result_in_b is NULL");
    if (segment_len < 0 || (segment_len > 0 && (segment_len % 4096 != 0 || n_max % segment_len != 0 || d_count)))
        return fail(PM_ERR_INVALID, "segment_len must be 0 or a multiple of 4096 dividing n_max (and d_count NULL)");
    *result_in_b = pm::rs_num_passes(begin_bit, end_bit) % 2;
    if (n_max == 0) return PM_OK;
    if (!keys_a || !keys_b || !vals_a || !vals_b) return fail(PM_ERR_INVALID, "key / value buffers are NULL");
    if (!scratch || scratch_bytes < static_cast<int64_t>(pm::rs_scratch_bytes(static_cast<size_t>(n_max))))
        return fail(PM_ERR_INVALID, "scratch too small: need " + std::to_string(pm::rs_scratch_bytes(static_cast<size_t>(n_max))) + " bytes");
    hipError_t h = key_bytes == 4
        ? pm::rs_sort_pairs<uint32_t>(static_cast<uint32_t*>(keys_a), static_cast<uint32_t*>(keys_b), vals_a, vals_b,
                                      static_cast<size_t>(n_max), d_count, begin_bit, end_bit, scratch, static_cast<hipStream_t>(stream),
                                      static_cast<size_t>(segment_len))
        : pm::rs_sort_pairs<uint64_t>(static_cast<uint64_t*>(keys_a), static_cast<uint64_t*>(keys_b), vals_a, vals_b,
                                      static_cast<size_t>(n_max), d_count, begin_bit, end_bit, scratch, static_cast<hipStream_t>(stream),
                                      static_cast<size_t>(segment_len));
    if (h != hipSuccess) return hip_fail(h, "pm_radix_sort_pairs");
    return PM_OK;
}
#endif

static int rowquant_args_ok(int64_t n_rows, int32_t dim, int32_t bitwidth) {
    if (n_rows < 0) return fail(PM_ERR_INVALID, "n_rows is negative");
    if (bitwidth != 16 && bitwidth != 8 && bitwidth != 4 && bitwidth != 2) return fail(PM_ERR_INVALID, "bitwidth must be 16, 8, 4 or 2");
    if (dim < 8 || dim % 8 != 0 || dim > 512) return fail(PM_ERR_UNSUPPORTED, "dim must be a multiple of 8 in [8, 512]");
    return PM_OK;
}

static uintptr_t rowquant_align(int32_t bitwidth) { return bitwidth == 16 ? 16 : static_cast<uintptr_t>(bitwidth); }

int64_t pm_rows_quantized_bytes(int64_t n_rows, int32_t dim, int32_t bitwidth) {
    const int rc = rowquant_args_ok(n_rows, dim, bitwidth);
    if (rc != PM_OK) return rc;
    return n_rows * pm::rows_quantized_row_bytes(dim, bitwidth);
}

int pm_rows_quantize(const float* src, int64_t n_rows, int32_t dim, int32_t bitwidth, void* dst, pm_stream_t stream) {
    const int rc = rowquant_args_ok(n_rows, dim, bitwidth);
    if (rc != PM_OK) return rc;
    if (n_rows == 0) return PM_OK;
    if (!src || !dst) return fail(PM_ERR_INVALID, "src / dst is NULL");
    if (reinterpret_cast<uintptr_t>(src) % 16 != 0 || reinterpret_cast<uintptr_t>(dst) % rowquant_align(bitwidth) != 0)
        return fail(PM_ERR_INVALID, "src must be 16-byte aligned, dst aligned to the format's word (16 / 8 / 4 / 2 bytes for 16 / 8 / 4 / 2 bits)");
    const hipError_t h = pm::launch_rows_quantize(src, n_rows, dim, bitwidth, dst, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_rows_quantize launch");
    return PM_OK;
}

int pm_rows_dequantize(const void* src, int64_t n_rows, int32_t dim, int32_t bitwidth, float* dst, pm_stream_t stream) {
    const int rc = rowquant_args_ok(n_rows, dim, bitwidth);
    if (rc != PM_OK) return rc;
    if (n_rows == 0) return PM_OK;
    if (!src || !dst) return fail(PM_ERR_INVALID, "src / dst is NULL");
    if (reinterpret_cast<uintptr_t>(dst) % 16 != 0 || reinterpret_cast<uintptr_t>(src) % rowquant_align(bitwidth) != 0)
        return fail(PM_ERR_INVALID, "dst must be 16-byte aligned, src aligned to the format's word (16 / 8 / 4 / 2 bytes for 16 / 8 / 4 / 2 bits)");
    const hipError_t h = pm::launch_rows_dequantize(src, n_rows, dim, bitwidth, dst, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_rows_dequantize launch");
    return PM_OK;
}

int pm_embbag_fwd(const pm_embbag_batch* op, float* out, pm_stream_t stream) {
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p, true);
    if (rc != PM_OK) return rc;
    if (p.bag_count == 0) return PM_OK;
    if (!out) return fail(PM_ERR_INVALID, "out is NULL");
    p.io = out;
    int unroll = g_unroll.load();
    if (unroll == 0) unroll = small_request_unroll(op, p);
    const hipError_t h = pm::launch_embbag_fwd(p, op->weight_dtype, op->max_dim, unroll, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_fwd launch");
    return PM_OK;
}

int pm_embbag_fwd_quantized(const pm_embbag_batch* op, void* out, int32_t bitwidth, pm_stream_t stream) {
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p, true);
    if (rc != PM_OK) return rc;
    if ((rc = rowquant_args_ok(0, op->max_dim, bitwidth)) != PM_OK) return rc;
    if (op->out_stride % op->max_dim != 0) return fail(PM_ERR_INVALID, "out_stride must be a whole number of max_dim-element rows");
    if (p.bag_count == 0) return PM_OK;
    if (!out) return fail(PM_ERR_INVALID, "out is NULL");
    if (reinterpret_cast<uintptr_t>(out) % 16 != 0) return fail(PM_ERR_INVALID, "out must be 16-byte aligned");
    if (p.flat_bags > 0) {   // the quantised burst lives in the bag-per-group kernel: back to its 8-bag tiles
        p.bags_per_block = p.stage_bags;
        p.flat_bags = 0;
    }
    if (!p.stage_out)
        return fail(PM_ERR_UNSUPPORTED, "quantised output needs the staged forward (fixed-pooling requests whose tile fits the staging "
                                        "buffer, staging not disabled): run pm_embbag_fwd and pm_rows_quantize instead");
    p.io = static_cast<float*>(out);
    p.out_bits = bitwidth;
    int unroll = g_unroll.load();
    if (unroll == 0) unroll = kDefaultUnroll;
    hipError_t h = pm::launch_embbag_fwd(p, op->weight_dtype, op->max_dim, unroll, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_fwd_quantized launch");
    return PM_OK;
}

int pm_embbag_fwd_split(const pm_embbag_batch* op, float* out, pm_stream_t stream) {
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p);
    if (rc != PM_OK) return rc;
    if (p.bag_count == 0) return PM_OK;
    if (!out) return fail(PM_ERR_INVALID, "out is NULL");
    p.io = out;
    hipError_t h = pm::launch_embbag_fwd_split(p, op->weight_dtype, op->max_dim, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_fwd_split launch");
    return PM_OK;
}

#ifdef PM_ALTERNATES
int pm_embbag_bwd(const pm_embbag_batch* op, const float* grad, void* const* dst_tables, int32_t dst_dtype,
                  float alpha, pm_stream_t stream) {
    pm::KParams p;
    int rc = make_params(op, dst_dtype, p);
    if (rc != PM_OK) return rc;
    if (p.bag_count == 0 || p.N == 0) return PM_OK;
    if (!grad || !dst_tables) return fail(PM_ERR_INVALID, "grad / dst_tables is NULL");
    p.io = const_cast<float*>(grad);
    p.tables = const_cast<const void* const*>(dst_tables);
    p.alpha = alpha;
    hipError_t h = pm::launch_embbag_bwd(p, dst_dtype, op->max_dim, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_bwd launch");
    return PM_OK;
}
#endif

static int sorted_args_ok(const pm_embbag_batch* op, int64_t max_rows) {
    if (max_rows < 1 || max_rows > (1LL << 31)) return fail(PM_ERR_INVALID, "max_rows must be in [1, 2^31]");
    if (op->num_indices >= (1LL << 32) || static_cast<int64_t>(op->batch) >= (1LL << 32))
        return fail(PM_ERR_UNSUPPORTED, "sorted backward needs num_indices and batch below 2^32");
#ifndef PM_ALTERNATES
    if (op->num_tables > pm::kSegSortMaxTables)
        return fail(PM_ERR_UNSUPPORTED, "sorted backward: at most " + std::to_string(pm::kSegSortMaxTables) +
                                            " tables per request (split the request by tables: the calls are independent)");
#endif
    return PM_OK;
}

int64_t pm_embbag_bwd_sorted_workspace(const pm_embbag_batch* op, int64_t max_rows) {
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p);
    if (rc != PM_OK) return rc;
    if ((rc = sorted_args_ok(op, max_rows)) != PM_OK) return rc;
    size_t bytes = 0;
    hipError_t h = pm::sorted_workspace_bytes(p, max_rows, op->max_dim, bytes);
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_bwd_sorted_workspace");
    return static_cast<int64_t>(bytes);
}

int pm_embbag_sort_indices(const pm_embbag_batch* op, int64_t max_rows, void* workspace, int64_t workspace_bytes,
                           pm_stream_t stream) {
    return pm_embbag_sort_indices_ex(op, max_rows, 1, workspace, workspace_bytes, stream);
}

static int sort_request(const pm_embbag_batch* op, int64_t max_rows, int32_t phases, void* workspace, int64_t workspace_bytes,
                        pm_stream_t stream, bool defer_ok) {
    if (phases != 1 && phases != 2) return fail(PM_ERR_INVALID, "phases must be 1 or 2");
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p);
    if (rc != PM_OK) return rc;
    if ((rc = sorted_args_ok(op, max_rows)) != PM_OK) return rc;
    if (p.N == 0) return PM_OK;
    size_t need = 0;
    hipError_t h = pm::sorted_workspace_bytes(p, max_rows, op->max_dim, need);
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_sort_indices");
    if (!workspace || workspace_bytes < static_cast<int64_t>(need))
        return fail(PM_ERR_INVALID, "workspace too small: need " + std::to_string(need) + " bytes");
    h = pm::sort_indices(p, max_rows, op->max_dim, op->fixed_pooling, phases, workspace, static_cast<hipStream_t>(stream), defer_ok);
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_sort_indices");
    return PM_OK;
}

int pm_embbag_sort_indices_ex(const pm_embbag_batch* op, int64_t max_rows, int32_t phases, void* workspace,
                              int64_t workspace_bytes, pm_stream_t stream) {
    return sort_request(op, max_rows, phases, workspace, workspace_bytes, stream, false);
}

// The fused calls validate what their APPLY half needs before the sort half launches anything (a NULL gradient or a bad
// dtype used to be found after the sort's kernels -- and, on the hybrid path, after a deferred plan -- had been issued).
static int fused_apply_args_ok(const pm_embbag_batch* op, const float* grad, void* const* tables, int32_t dtype) {
    if (!op) return fail(PM_ERR_INVALID, "op is NULL");
    if (!dtype_is_weight(dtype)) return fail(PM_ERR_INVALID, "weight/dst dtype must be PM_F32, PM_BF16 or PM_F16");
    if (!grad || !tables) return fail(PM_ERR_INVALID, "grad / dst_tables is NULL");
    return PM_OK;
}

int pm_embbag_bwd_fused(const pm_embbag_batch* op, const float* grad, void* const* dst_tables, int32_t dst_dtype, float alpha,
                        int64_t max_rows, void* workspace, int64_t workspace_bytes, pm_stream_t stream) {
    if (op && (op->num_indices == 0 || op->bag_count == 0)) return PM_OK;
    int rc = fused_apply_args_ok(op, grad, dst_tables, dst_dtype);
    if (rc != PM_OK) return rc;
    {
        pm::KParams chk;                            // the apply builds its parameters for the DESTINATION dtype: its row-width rule too
        if ((rc = make_params(op, dst_dtype, chk)) != PM_OK) return rc;
    }
    rc = sort_request(op, max_rows, 1, workspace, workspace_bytes, stream, true);
    if (rc != PM_OK) return rc;
    return pm_embbag_bwd_sorted(op, grad, dst_tables, dst_dtype, alpha, max_rows, workspace, workspace_bytes, stream);
}

int pm_embbag_bwd_fused_adagrad(const pm_embbag_batch* op, const float* grad, void* const* tables, int32_t table_dtype,
                                float* const* momentum, const pm_rowwise_adagrad* opt, int64_t max_rows, void* workspace,
                                int64_t workspace_bytes, pm_stream_t stream) {
    if (op && (op->num_indices == 0 || op->bag_count == 0)) return PM_OK;
    int rc = fused_apply_args_ok(op, grad, tables, table_dtype);
    if (rc != PM_OK) return rc;
    {
        pm::KParams chk;
        if ((rc = make_params(op, table_dtype, chk)) != PM_OK) return rc;
    }
    if (!opt) return fail(PM_ERR_INVALID, "optimizer options are NULL");
    if (opt->weight_decay_mode != PM_WD_NONE && opt->weight_decay_mode != PM_WD_L2 && opt->weight_decay_mode != PM_WD_DECOUPLE)
        return fail(PM_ERR_INVALID, "weight_decay_mode must be PM_WD_NONE, PM_WD_L2 or PM_WD_DECOUPLE");
    if (!momentum) return fail(PM_ERR_INVALID, "grad / tables / momentum is NULL");
    if (op->max_dim > 64 * ((table_dtype == PM_F32) ? 4 : 8))
        return fail(PM_ERR_UNSUPPORTED, "row-wise Adagrad needs max_dim <= " + std::to_string(64 * ((table_dtype == PM_F32) ? 4 : 8)) + " for this table dtype");
    rc = sort_request(op, max_rows, 1, workspace, workspace_bytes, stream, true);
    if (rc != PM_OK) return rc;
    return pm_embbag_bwd_sorted_adagrad_ex(op, grad, tables, table_dtype, momentum, opt, max_rows, workspace, workspace_bytes, stream);
}

int pm_embbag_sort_plan(const pm_embbag_batch* op, int64_t max_rows, int32_t phases, char* out, int32_t out_bytes) {
    if (phases != 1 && phases != 2) return fail(PM_ERR_INVALID, "phases must be 1 or 2");
    if (!out || out_bytes < 1) return fail(PM_ERR_INVALID, "out buffer is NULL / empty");
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p);
    if (rc != PM_OK) return rc;
    if ((rc = sorted_args_ok(op, max_rows)) != PM_OK) return rc;
    const std::string d = pm::sort_plan_describe(p, max_rows, op->fixed_pooling, phases);
    snprintf(out, static_cast<size_t>(out_bytes), "%s", d.c_str());
    return PM_OK;
}

int pm_embbag_sorted_pairs(const pm_embbag_batch* op, int64_t max_rows, const void* workspace, const void** keys,
                           const uint32_t** vals, const uint32_t** d_count, int32_t* key_bytes, int32_t* tshift) {
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p);
    if (rc != PM_OK) return rc;
    if ((rc = sorted_args_ok(op, max_rows)) != PM_OK) return rc;
    if (!workspace || !keys || !vals || !d_count || !key_bytes || !tshift) return fail(PM_ERR_INVALID, "NULL argument");
    int kb = 0, ts = 0;
    const int pi = pm::sorted_pairs_info(p, max_rows, op->max_dim, workspace, keys, vals, d_count, &kb, &ts);
    if (pi == 2)
        return fail(PM_ERR_INVALID, "the last sort on this workspace was deferred into its apply (pm_embbag_bwd_fused*, hybrid backward) and no "
                                    "apply has been issued yet: there are no sorted pairs to look at");
    if (pi != 0) return fail(PM_ERR_INVALID, "no sort has been recorded for this workspace");
    *key_bytes = kb;
    *tshift = ts;
    return PM_OK;
}

int pm_embbag_bwd_sorted(const pm_embbag_batch* op, const float* grad, void* const* dst_tables, int32_t dst_dtype,
                         float alpha, int64_t max_rows, const void* workspace, int64_t workspace_bytes,
                         pm_stream_t stream) {
    pm::KParams p;
    int rc = make_params(op, dst_dtype, p);
    if (rc != PM_OK) return rc;
    if ((rc = sorted_args_ok(op, max_rows)) != PM_OK) return rc;
    if (p.N == 0 || p.bag_count == 0) return PM_OK;
    if (!grad || !dst_tables) return fail(PM_ERR_INVALID, "grad / dst_tables is NULL");
    size_t need = 0;
    hipError_t h = pm::sorted_workspace_bytes(p, max_rows, op->max_dim, need);
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_bwd_sorted");
    if (!workspace || workspace_bytes < static_cast<int64_t>(need))
        return fail(PM_ERR_INVALID, "workspace too small: need " + std::to_string(need) + " bytes");
    if (pm::bwd_sorted_plan_check(p, max_rows, workspace, false) != 0)
        return fail(PM_ERR_INVALID, "pm_embbag_sort_indices has not been called for this request on this workspace (same indices / offsets pointers, batch, bag slice and weights as the sort's)");
    p.io = const_cast<float*>(grad);
    p.tables = const_cast<const void* const*>(dst_tables);
    p.alpha = alpha;
    if (g_nt_loads.load() < 0) p.nt_loads = kDefaultRowPolicy;
    h = pm::bwd_sorted_apply(p, max_rows, dst_dtype, op->max_dim, workspace, nullptr, nullptr,
                             static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_bwd_sorted launch");
    return PM_OK;
}

int pm_embbag_bwd_sorted_adagrad_ex(const pm_embbag_batch* op, const float* grad, void* const* tables, int32_t table_dtype,
                                    float* const* momentum, const pm_rowwise_adagrad* opt, int64_t max_rows,
                                    const void* workspace, int64_t workspace_bytes, pm_stream_t stream) {
    pm::KParams p;
    int rc = make_params(op, table_dtype, p);
    if (rc != PM_OK) return rc;
    if ((rc = sorted_args_ok(op, max_rows)) != PM_OK) return rc;
    if (!opt) return fail(PM_ERR_INVALID, "optimizer options are NULL");
    if (opt->weight_decay_mode != PM_WD_NONE && opt->weight_decay_mode != PM_WD_L2 && opt->weight_decay_mode != PM_WD_DECOUPLE)
        return fail(PM_ERR_INVALID, "weight_decay_mode must be PM_WD_NONE, PM_WD_L2 or PM_WD_DECOUPLE");
    if (p.N == 0 || p.bag_count == 0) return PM_OK;
    if (!grad || !tables || !momentum) return fail(PM_ERR_INVALID, "grad / tables / momentum is NULL");
    const int vec = (table_dtype == PM_F32) ? 4 : 8;
    if (op->max_dim > 64 * vec)
        return fail(PM_ERR_UNSUPPORTED, "row-wise Adagrad needs max_dim <= " + std::to_string(64 * vec) + " for this table dtype");
    size_t need = 0;
    hipError_t h = pm::sorted_workspace_bytes(p, max_rows, op->max_dim, need);
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_bwd_sorted_adagrad");
    if (!workspace || workspace_bytes < static_cast<int64_t>(need))
        return fail(PM_ERR_INVALID, "workspace too small: need " + std::to_string(need) + " bytes");
    {
        const int pc = pm::bwd_sorted_plan_check(p, max_rows, workspace, true);
        if (pc == 2)
            return fail(PM_ERR_INVALID, "the request was sorted for a two-phase scatter-add apply; row-wise Adagrad needs "
                                        "pm_embbag_sort_indices (phases = 1)");
        if (pc != 0)
            return fail(PM_ERR_INVALID, "pm_embbag_sort_indices has not been called for this request on this workspace (same indices / offsets pointers, batch, bag slice and weights as the sort's)");
    }
    p.io = const_cast<float*>(grad);
    p.tables = const_cast<const void* const*>(tables);
    p.alpha = 1.0f;
    if (g_nt_loads.load() < 0) p.nt_loads = kDefaultRowPolicy;
    h = pm::bwd_sorted_apply(p, max_rows, table_dtype, op->max_dim, workspace, momentum, opt, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_bwd_sorted_adagrad launch");
    return PM_OK;
}

int pm_embbag_bwd_sorted_adagrad(const pm_embbag_batch* op, const float* grad, void* const* tables, int32_t table_dtype,
                                 float* const* momentum, float lr, float eps, int64_t max_rows, const void* workspace,
                                 int64_t workspace_bytes, pm_stream_t stream) {
    pm_rowwise_adagrad opt = {lr, eps, 0.0f, PM_WD_NONE, 0, 0, 0};
    return pm_embbag_bwd_sorted_adagrad_ex(op, grad, tables, table_dtype, momentum, &opt, max_rows, workspace,
                                           workspace_bytes, stream);
}

int pm_dlrm_regroup(const int64_t* lengths, const int64_t* indices, int32_t world_size, int32_t num_tables,
                    int64_t batch, int64_t* out_indices, int64_t* out_offsets, int64_t* scratch, pm_stream_t stream) {
    if (world_size < 1 || num_tables < 1 || batch < 0) return fail(PM_ERR_INVALID, "world_size / num_tables / batch");
    if (static_cast<int64_t>(world_size) * num_tables > 4096) return fail(PM_ERR_UNSUPPORTED, "world_size * num_tables > 4096");
    if (!lengths || !out_offsets || !scratch) return fail(PM_ERR_INVALID, "lengths / out_offsets / scratch is NULL");
    hipError_t h = pm::launch_dlrm_regroup(lengths, indices, world_size, num_tables, batch, out_indices, out_offsets,
                                           scratch, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_dlrm_regroup launch");
    return PM_OK;
}

int pm_embbag_check_ex(const pm_embbag_batch* op, int32_t flags, int32_t* d_error_count, pm_stream_t stream) {
    pm::KParams p;
    int rc = make_params(op, op ? op->weight_dtype : -1, p);
    if (rc != PM_OK) return rc;
    if (!d_error_count) return fail(PM_ERR_INVALID, "d_error_count is NULL");
    if (flags & ~PM_CHECK_UNIFORM_DIMS) return fail(PM_ERR_INVALID, "unknown check flag");
    hipError_t h = pm::launch_embbag_check(p, d_error_count, op->weight_dtype == PM_F32 ? 4 : 8, op->max_dim, op->fixed_pooling,
                                           (flags & PM_CHECK_UNIFORM_DIMS) ? 1 : 0, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_embbag_check launch");
    return PM_OK;
}

int pm_embbag_check(const pm_embbag_batch* op, int32_t* d_error_count, pm_stream_t stream) {
    return pm_embbag_check_ex(op, 0, d_error_count, stream);
}

int pm_fill_random(void* dst, int64_t count, int32_t dtype, int32_t dist, float lo, float hi, uint64_t seed,
                   pm_stream_t stream) {
    if (count < 0) return fail(PM_ERR_INVALID, "negative count");
    if (count > 0 && !dst) return fail(PM_ERR_INVALID, "dst is NULL");
    if (!dtype_is_weight(dtype)) return fail(PM_ERR_INVALID, "dtype must be PM_F32, PM_BF16 or PM_F16");
    if (dist != 0 && dist != 1) return fail(PM_ERR_INVALID, "dist must be 0 (uniform) or 1 (normal)");
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) != 0) return fail(PM_ERR_INVALID, "dst must be 16-byte aligned");
    hipError_t h = pm::launch_fill_random(dst, count, dtype, dist, lo, hi, seed, static_cast<hipStream_t>(stream));
    if (h != hipSuccess) return hip_fail(h, "pm_fill_random launch");
    return PM_OK;
}

}  // extern "C"
