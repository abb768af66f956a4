// param_amd/csrc/embbag_bwd_sorted.hip -- deterministic EmbeddingBag backward (no atomics).
//
//     dst_t[indices[j], :] += alpha * psw[j] * grad(t, bag(j))[:]
//
// Why not atomics: on MI355X a device-scope float atomic is a fabric transaction per dword;
// the atomic kernel (embbag_bwd.hip) tops out at ~77 G atomic-dwords/s = 0.6 G lookups/s at
// D=128 (profiles/, sweep r1a), 6-8 % of the HBM roofline.  This path instead
//   1. builds one (table,row) key and one bag value per lookup          (build_keys_kernel)
//   2. sorts the pairs with a stable LSD radix sort by the ROW bits only (radix_sort.hip: own kernels, 3 passes of 8 bits
//      for 10 M-row tables).  The request is table-major, so after a stable sort by row the lookups of one (table, row)
//      are still contiguous and in lookup order -- the order is (row, table, position), which is all step 3 needs: it
//      finds runs by key equality.  (Sorting the table bits too would be a fourth pass for nothing.)  (That was round 2; the
//      product library sorts with seg_sort.hip -- per-table segments established on the device.  Round 2's sort and rocPRIM's
//      radix_sort_pairs are measured alternatives and cross-checks of the ALTERNATES build only: make alt, -DPM_ALTERNATES.)
//   3. streams the sorted pairs: every run of equal keys is owned by ONE lane group, which
//      reads the destination row once, adds the run's gradient rows in sorted (= original
//      index) order in fp32 registers and writes the row back once      (bwd_sorted_kernel)
// so each touched row costs one HBM read + one HBM write (the algorithmic 2*D*e bytes), the
// gradient rows are re-read from L2 / Infinity Cache, and there is no atomic and no race.
// The sort is stable, so within a row contributions are added in increasing lookup position:
// the result is bit-identical to the sequential CPU oracle (oracle/embbag_oracle.c) and
// run-to-run deterministic -- what torch's CUDA dense backward obtains by sort + segmented
// reduce (aten::_embedding_bag_dense_backward) and fbgemm's TBE backward by its sorted
// linear indices (reference call sites: pytorch_dist_backend.py:854-857,
// split_table_batched_embeddings_ops.py:318-324).
//
// Step 1+2 depend only on the indices, not on the gradient: pm_embbag_sort_indices() can run
// on a side stream under the forward pass; pm_embbag_bwd_sorted() is step 3.
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#ifdef PM_ALTERNATES
#include <rocprim/device/device_radix_sort.hpp>
#endif

#include "bwd_sorted_apply.h"

namespace pm {
namespace {

constexpr int kDefaultSortMode = 0;
// (kSortTile, SortedParams, the destination types, the kernels and their launchers: bwd_sorted_apply.h / _impl.inc)

// ---------------------------------------------------------------------------------------------
#ifdef PM_ALTERNATES      // (round 2's sort only: the segmented sort forms its keys itself)
// step 1: keys / values, same tiling and LDS offset staging as the forward
template <typename K, bool WEIGHTED>
__global__ void __launch_bounds__(kBlock) build_keys_kernel(const KParams p, K* keys, uint32_t* vals,
                                                            uint32_t* bag_of, int rbits, int tshift, int kbits,
                                                            int64_t phase_bags, int64_t slice_begin, int64_t slice_end) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int t, tile;
    block_to_tile(p, t, tile);
    if (t >= p.T) return;
    const int64_t bag0 = static_cast<int64_t>(tile) * p.bags_per_block;  // p.bag_begin == 0 here: whole batch
    const int64_t left = p.B - bag0;
    const int nb = left < p.bags_per_block ? static_cast<int>(left) : p.bags_per_block;
    const int64_t g0 = static_cast<int64_t>(t) * p.B + bag0;
    int64_t* s_off = reinterpret_cast<int64_t*>(smem);
    for (int i = threadIdx.x; i <= nb; i += kBlock) s_off[i] = bag_start_or_end(p, g0 + i);
    __syncthreads();
    const int64_t base = s_off[0];
    const int64_t end = s_off[nb];
    const K pad = static_cast<K>(1) << kbits;
    for (int64_t j = base + threadIdx.x; j < end; j += kBlock) {
        // bag of lookup j: largest b with s_off[b] <= j (binary search over the LDS offsets)
        int lo = 0, hi = nb;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= j) lo = mid; else hi = mid;
        }
        const int64_t bag = bag0 + lo;
        const K row = static_cast<K>(load_index(p.indices, j, p.idx64));
        const bool in_slice = bag >= slice_begin && bag < slice_end;
        // (table, bag phase, row): phase = which run of phase_bags consecutive bags the lookup belongs to (0 everywhere when
        // the apply runs in one phase); the phase sits between table and row so that a (table, phase) segment sorts by row
        const K phase = phase_bags > 0 ? static_cast<K>(bag / phase_bags) : static_cast<K>(0);
        keys[j] = in_slice ? ((static_cast<K>(t) << tshift) | (phase << rbits) | row) : pad;
        if (WEIGHTED) {
            vals[j] = static_cast<uint32_t>(j);
            bag_of[j] = static_cast<uint32_t>(bag);
        } else {
            vals[j] = static_cast<uint32_t>(bag);
        }
    }
}
#endif



// ---------------------------------------------------------------------------------------------
// workspace layout (shared by the sort and apply entry points)
struct SortWs {
    char* keys_a;
    char* keys_b;
    uint32_t* vals_a;
    uint32_t* vals_b;
    uint32_t* bag_of;
    ChunkRec* recs;
    uint32_t* fix_list;      // chunk ids the apply's main kernel hands to its fix-up kernel
    uint32_t* fix_ctl;       // two list lengths and two generation words (bwd_sorted_apply.h): zeroed by every sort
    float* partials;
    void* temp;
    size_t temp_bytes;
    size_t total;
};

// chunks of the apply kernels: kSortTile / (kBlock / G) positions each; sized for the smallest
// chunk any destination dtype can select for this max_dim (16-bit destinations: 8 elements/lane)
// sorted positions per workgroup of the apply kernels: 512 (256 below 768 K lookups, where 512 would leave CUs idle).  Measured
// with the kernels' load batches real and tile windows that keep short runs whole (round 3, visit v49; apply ms uniform / Zipf
// on 48 x 10 M x 128 fp32, Criteo backward us Zipf / uniform):  256: 1.452 / 0.945, 426 / 488;  512: 1.468 / 0.835, 403 / 470;
// 1024: 1.496 / 0.815, 394 / 481.  Smaller tiles balance the tail of a launch better; larger ones cut a long run (a Zipf head,
// a 3-row table) into fewer pieces for the fix-up kernel.  PARAM_AMD_BWD_TILE overrides (256 / 512 / 1024).
inline int apply_tile(int64_t n) {
    // read ONCE per process: the value also sizes the workspace (max_chunks), so the workspace query, the sort and the apply of a
    // request must see the same one
    static const int env = [] { const char* e = getenv("PARAM_AMD_BWD_TILE"); return e ? atoi(e) : 0; }();
    if (env == 256 || env == 512 || env == 1024) return env;
    return n < (static_cast<int64_t>(3) << 18) ? 256 : 512;
}

inline int64_t max_chunks(int64_t n, int max_dim) {
    const int g = group_lanes(max_dim, 8);
    const int c = apply_tile(n) / (kBlock / g);
    return (n + c - 1) / c + 1;
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

#ifdef PM_ALTERNATES
template <typename K>
hipError_t rocprim_temp_bytes(int64_t n, int kbits_sort, size_t& bytes) {
    bytes = 0;
    K* kn = nullptr;
    uint32_t* vn = nullptr;
    return rocprim::radix_sort_pairs(nullptr, bytes, kn, kn, vn, vn, static_cast<size_t>(n), 0u,
                                     static_cast<unsigned>(kbits_sort), hipStream_t(0));
}
#endif

// Backward tuning knobs (pm_set_backward_tuning; -1 = default, which the environment can override once):
//   sort_impl  0 own radix sort (radix_sort.hip), 1 rocPRIM radix_sort_pairs              PARAM_AMD_SORT=rocprim
//   order      1 (table, [phase,] row, position) -- default --, 0 (row, table, position):   PARAM_AMD_SORT_ORDER=row
//              only the row bits are sorted (one pass fewer; the apply kernel then runs 5 % slower and cannot be XCD-affine)
//   xcd        1 (default): XCD-affine tile mapping of the apply kernel where the layout allows  PARAM_AMD_BWD_XCD=0
std::atomic<int> g_sort_impl{-1}, g_sort_order{-1}, g_bwd_xcd{-1}, g_max_phases{-1};
// (called from knob(): each knob's environment default is looked up the first time the knob is read and stored in the knob)
int env_is(const char* name, const char* value) {
    const char* e = getenv(name);
    return (e && std::string(e) == value) ? 1 : 0;
}
// a knob left at -1 takes its default from the environment, looked up ONCE, when the knob is first read (and again only after a
// pm_set_* call has put it back to -1): never on a launch path
template <typename F>
int knob(std::atomic<int>& k, F env_default) {
    int v = k.load();
    if (v < 0) {
        v = env_default();
        k.store(v);
    }
    return v;
}
// sort_impl: 0 (default) the segmented sort of round 3 (seg_sort.hip: per-table segments established on the device);
//            1 rocPRIM radix_sort_pairs (PARAM_AMD_SORT=rocprim); 2 round 2's own LSD sort with host-side plans
//            (PARAM_AMD_SORT=legacy) -- both kept as measured alternatives and as independent checks of the new path
//            (the alternates build only; the product library has the segmented sort and nothing else)
#ifdef PM_ALTERNATES
int sort_impl_knob() { return knob(g_sort_impl, [] { return env_is("PARAM_AMD_SORT", "rocprim") ? 1 : env_is("PARAM_AMD_SORT", "legacy") ? 2 : 0; }); }
#else
int sort_impl_knob() { return 0; }
#endif
bool use_rocprim_sort() { return sort_impl_knob() == 1; }
// how the segmented sort orders a table's pairs (pm_set_sort_tuning, PARAM_AMD_SORT_MODE): 0 LSD passes over all row bits
// (ascending rows; one kernel per pass, tiles learn their prefixes from their predecessors in flight), 1 one partition pass on
// the low row digit + bucket-local sort in LDS, 2 the same on the top digit, 3 = 0 with histogram / scan / scatter kernels per pass
std::atomic<int> g_sort_mode{-1};
int sort_mode_knob() {
    int v = g_sort_mode.load();
    if (v < 0) {
        const char* e = getenv("PARAM_AMD_SORT_MODE");
        v = (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : kDefaultSortMode;
        g_sort_mode.store(v);
    }
    return v;
}
// round 2's sort (sort_impl 2, a measured alternative and cross-check) only: the default segmented sort returns from make_plan before
// this is looked at, so no default launch path reads the environment; tests flip it between two plans of one process
bool fused_keys_allowed() { return !env_is("PARAM_AMD_SORT_FUSED_KEYS", "0"); }
bool table_major_order() { return knob(g_sort_order, [] { return env_is("PARAM_AMD_SORT_ORDER", "row") ? 0 : 1; }) == 1; }
// hybrid backward (pm_set_hybrid_tuning; common.h "Hybrid backward"):
//   enable   0 off; 1 (default) on: every table is classified on the device at every sort, from the request alone; 2 every
//            structurally eligible table takes the hybrid path whatever its indices look like (tests)   PARAM_AMD_BWD_HYBRID=0..2
//   spin_cap look-back polls before a walk stops waiting and counts its predecessor's digits itself (0 = default 2^12; tests: 1)
// (Round 4 also built the rest of the sort + the sorted apply of the flagged lookups on a second, library-owned stream beside
//  the bag-major apply -- disjoint rows -- and measured it: the small kernels do run concurrently, and starve: the 54 us emit
//  pass took 1.26 ms beside the chip-filling kernel, which itself went from 1.45 to 1.72 ms.  Removed; profiles/r04_*.)
std::atomic<int> g_hyb_enable{-1};
std::atomic<uint32_t> g_spin_cap{0};
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && e[0] >= '0' && e[0] <= '9') ? atoi(e) : dflt;
}
int hyb_enable_knob() { return knob(g_hyb_enable, [] { return env_int("PARAM_AMD_BWD_HYBRID", 1); }); }
// the hybrid tables' left-overs finished in LDS by hyb_rest_kernel (round 6; pm_set_hybrid_rest): 1 (default) on, 0 every list goes
// through the key sort and the sorted apply as in round 5 (A/B runs, cross-check)                     PARAM_AMD_HYB_REST=0
// (Round 6 also ran hyb_rest_kernel on a library-owned stream BESIDE the key sort's launches -- it touches the stage and its own
//  tables' rows only; forked by an event after the staging kernel, joined by one after the chain's last kernel -- to hide the eight
//  launches that find nothing to do (4.8 us each) under it.  Same box, modes taking turns, fused backward ms: fp32 uniform 1.549 (round-5
//  route) / 1.534 (LDS kernel on the caller's stream) / 1.544 (side stream); bf16 1.162 / 1.150 / 1.155; and the Zipf step, which only
//  pays for the fork and join, 0.957 / 0.957 / 0.972-0.981: two cross-stream event hand-offs cost more than the overlap returns.
//  Removed; profiles/r06_rest_kernel_ab.md.)
std::atomic<int> g_hyb_rest{-1};
std::atomic<int> g_hyb_min_tiles{-1};      // pm_set_hybrid_min_tiles: bag-major workgroups from which the hybrid path is offered (-1: kHybMinTiles)
int hyb_rest_knob() { return knob(g_hyb_rest, [] { return env_int("PARAM_AMD_HYB_REST", 1) != 0 ? 1 : 0; }); }
bool want_xcd() { return knob(g_bwd_xcd, [] { return env_is("PARAM_AMD_BWD_XCD", "0") ? 0 : 1; }) == 1; }
//   max_phases 1 (default): one apply launch; 2: a phases = 2 sort lays a fixed-pooling request out for the two-phase
//              apply (measured at benchmark size: uniform indices 1.60 -> 1.58 ms, Zipf 0.97 -> 1.12 ms: rows looked up in
//              both bag halves are read and written twice, and halving the gradient working set does not make it stay in
//              L2 -- 1 KB of row traffic streams through for every 512 B gradient row)              PARAM_AMD_BWD_PHASES=2
int max_phases() { return knob(g_max_phases, [] { return env_is("PARAM_AMD_BWD_PHASES", "2") ? 2 : 1; }); }

hipError_t ws_layout(void* base, int64_t n, int T, int key_bytes, int kbits_sort, bool weighted, int max_dim, SortWs& ws) {
    size_t tb = 0;
#ifdef PM_ALTERNATES
    hipError_t rc = key_bytes == 4 ? rocprim_temp_bytes<uint32_t>(n, kbits_sort > 32 ? 32 : kbits_sort, tb)
                                   : rocprim_temp_bytes<uint64_t>(n, kbits_sort, tb);
    if (rc != hipSuccess) return rc;
    const size_t own = rs_scratch_bytes(static_cast<size_t>(n));
    if (own > tb) tb = own;
#else
    if (T > kSegSortMaxTables) return hipErrorInvalidValue;      // (capi.hip refuses such a request with a message before it gets here)
#endif
    if (T <= kSegSortMaxTables) {
        const size_t seg = seg_sort_scratch_bytes(static_cast<size_t>(n), T, !weighted);
        if (seg > tb) tb = seg;
    }
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = p ? p + off : nullptr; off += align256(bytes); return q; };
    ws.keys_a = take(static_cast<size_t>(n) * key_bytes);
    ws.keys_b = take(static_cast<size_t>(n) * key_bytes);
    ws.vals_a = reinterpret_cast<uint32_t*>(take(static_cast<size_t>(n) * 4));
    ws.vals_b = reinterpret_cast<uint32_t*>(take(static_cast<size_t>(n) * 4));
    ws.bag_of = reinterpret_cast<uint32_t*>(take(weighted ? static_cast<size_t>(n) * 4 : 0));
    const size_t nch = static_cast<size_t>(max_chunks(n, max_dim));
    ws.recs = reinterpret_cast<ChunkRec*>(take(nch * sizeof(ChunkRec)));
    ws.fix_list = reinterpret_cast<uint32_t*>(take(nch * sizeof(uint32_t)));
    ws.fix_ctl = reinterpret_cast<uint32_t*>(take(4 * sizeof(uint32_t)));
    ws.partials = reinterpret_cast<float*>(take(nch * 2 * static_cast<size_t>(max_dim) * sizeof(float)));
    ws.temp = take(tb);
    ws.temp_bytes = tb;
    ws.total = off;
    return hipSuccess;
}

inline int bits_for(int64_t n_values) {  // bits needed to represent 0 .. n_values-1
    int b = 0;
    while ((static_cast<int64_t>(1) << b) < n_values) ++b;
    return b;
}

// Everything the sort and the apply have to agree on, derived in ONE place from the request, the knobs and the number
// of bag phases asked for.  pm_embbag_sort_indices records the plan it sorted under (keyed by the workspace); the apply
// entry points use that record.
//
// key = (table << tshift) | (phase << rbits) | row, padding keys (batch slices only) = 1 << kbits.
//   fixed pooling (every bag L lookups, op->fixed_pooling = L) and a whole-batch request make the table-major request a
//   sequence of T * H equal segments of seg_len = (B / H) * L lookups, H = bag phases.  If seg_len is a multiple of the
//   sort tile the own sort orders every segment on its own by the ROW bits only (3 passes of 8 bits for 10 M rows: the
//   table and phase bits need no pass); if it is a multiple of the apply tile the apply kernel can run XCD-affine and
//   in H launches.  Everything else (ragged bags, slices, odd sizes) sorts all key bits globally and applies in one launch.
struct SortPlan {
    int key_bytes, rbits, hbits, tshift, kbits;
    bool sliced, weighted, rocprim, segmented, in_b, xcd, fused_keys;
    int H;
    int64_t n, seg_len, phase_bags;
    int sort_end_bit;
    int32_t seg_tiles;   // apply tiles (kSortTile) per segment, 0 = no segment structure
    int32_t T;
    bool v2;             // sorted by seg_sort.hip: segments, pooling and the pair count live on the device
    int mode;            // seg_sort mode (0 / 3 LSD passes, 1 / 2 partition + bucket-local sort)
    int hyb;             // hybrid backward: 0 not launched for this sort, else the `allow` value its kernels ran with (part B of the
                         // sort then runs inside the apply call, after the bag-major kernel)
    bool applied;        // an apply has been issued for this sort (a deferred sort's pairs exist only then)
    // what the sort was issued for: the apply must follow with the same request on the same workspace
    const void* indices;
    const void* offsets;
    int64_t B, bag_begin, bag_count;
    uint64_t stamp;
};

SortPlan make_plan(const KParams& p, int64_t max_rows, int64_t fixed_pooling, int phases) {
    SortPlan g;
    g.T = p.T;
    g.n = p.N;
    g.rbits = bits_for(max_rows);
    g.sliced = !(p.bag_begin == 0 && p.bag_count == p.B);
    g.weighted = p.psw != nullptr;
    g.rocprim = use_rocprim_sort();
    g.indices = p.indices;
    g.offsets = p.offsets;
    g.B = p.B;
    g.bag_begin = p.bag_begin;
    g.bag_count = p.bag_count;
    g.stamp = 0;
    g.v2 = sort_impl_knob() == 0 && p.T <= kSegSortMaxTables;
    g.mode = sort_mode_knob();
    g.hyb = 0;
    g.applied = false;
    if (g.v2) {
        // one plan for every request: the device establishes segments, per-table pooling and (for slices) the pair count
        g.H = 1;
        g.hbits = 0;
        g.tshift = g.rbits;
        g.kbits = g.tshift + bits_for(p.T);
        g.key_bytes = (g.kbits + 1 <= 32) ? 4 : 8;      // one spare bit: the apply kernel's "no neighbour" sentinel
        g.phase_bags = 0;
        g.seg_len = 0;
        g.seg_tiles = 0;
        g.segmented = true;
        g.xcd = want_xcd();
        g.sort_end_bit = g.rbits;
        g.in_b = seg_sort_result_in_b(g.mode, g.rbits);
        g.fused_keys = !g.weighted;
        return g;
    }
#ifdef PM_ALTERNATES
    const bool fixed = fixed_pooling > 0 && !g.sliced && p.T >= 1 && p.B > 0 &&
                       fixed_pooling * p.B * static_cast<int64_t>(p.T) == p.N;
    g.H = 1;
    if (phases == 2 && max_phases() >= 2 && fixed && table_major_order() && p.B % 2 == 0 && ((p.B / 2) * fixed_pooling) % kSortTile == 0) g.H = 2;
    g.hbits = g.H == 2 ? 1 : 0;
    g.tshift = g.rbits + g.hbits;
    g.kbits = g.tshift + bits_for(p.T);
    g.key_bytes = (g.kbits + 1 <= 32) ? 4 : 8;
    g.phase_bags = g.H == 2 ? p.B / 2 : 0;
    g.seg_len = fixed ? (p.B / g.H) * fixed_pooling : 0;
    g.seg_tiles = (fixed && table_major_order() && g.seg_len % kSortTile == 0) ? static_cast<int32_t>(g.seg_len / kSortTile) : 0;
    // (one workgroup walks a segment's tile counts: beyond a few thousand tiles per segment the global scan is the faster one)
    g.segmented = fixed && table_major_order() && !g.rocprim && g.seg_len > 0 && g.seg_len % 4096 == 0 && g.seg_len / 4096 <= 4096;
    g.xcd = want_xcd() && g.seg_tiles > 0 && p.T > 1;
    if (g.sliced) g.sort_end_bit = g.kbits + 1;                  // padding keys must end up last
    else if (g.segmented) g.sort_end_bit = g.rbits;              // per (table, phase) segment: rows only
    else g.sort_end_bit = table_major_order() ? g.kbits : g.rbits;
    g.in_b = g.rocprim || (rs_num_passes(0, g.sort_end_bit) % 2 == 1);
    g.fused_keys = g.segmented && g.H == 1 && !g.weighted && !g.sliced && g.sort_end_bit > 0 && fused_keys_allowed();
#else
    // (more than kSegSortMaxTables tables: refused by capi.hip; the fields below keep the record well-formed)
    g.H = 1;
    g.hbits = 0;
    g.tshift = g.rbits;
    g.kbits = g.tshift + bits_for(p.T);
    g.key_bytes = (g.kbits + 1 <= 32) ? 4 : 8;
    g.phase_bags = 0;
    g.seg_len = 0;
    g.seg_tiles = 0;
    g.segmented = false;
    g.xcd = false;
    g.sort_end_bit = g.kbits;
    g.in_b = false;
    g.fused_keys = false;
    (void)fixed_pooling;
    (void)phases;
#endif
    return g;
}

std::mutex g_plan_mutex;
std::unordered_map<const void*, SortPlan> g_plans;   // workspace -> the plan of the last sort issued on it
uint64_t g_plan_stamp = 0;

template <typename K>
SegSortRequest seg_request(const KParams& p, const SortPlan& g, SortWs& ws) {
    SegSortRequest rq;
    rq.indices = p.indices;
    rq.offsets = p.offsets;
    rq.rows = p.rows;
    rq.idx64 = p.idx64;
    rq.T = p.T;
    rq.B = p.B;
    rq.N = p.N;
    rq.bag_begin = p.bag_begin;
    rq.bag_count = p.bag_count;
    rq.tshift = g.tshift;
    rq.rbits_max = g.rbits;
    rq.weighted = g.weighted;
    rq.zero4 = ws.fix_ctl;
    rq.hyb.allow = g.hyb;
    rq.hyb.slices = hyb_slices(p.N, p.T);
    rq.queue_a = ws.vals_a;
    rq.queue_b = ws.vals_b;
    rq.spin_cap = g_spin_cap.load();
    return rq;
}

template <typename K>
hipError_t sort_impl(const KParams& p, const SortPlan& g, SortWs& ws, hipStream_t stream) {
    K* ka = reinterpret_cast<K*>(ws.keys_a);
    K* kb = reinterpret_cast<K*>(ws.keys_b);
    if (g.v2) {
        const SegSortRequest rq = seg_request<K>(p, g, ws);
        const hipError_t rc = seg_sort_part_a<K>(rq, ws.temp, stream);
        if (rc != hipSuccess || g.hyb) return rc;      // hybrid: the rest of the sort follows the bag-major kernel, in the apply call
        return seg_sort_part_b<K>(rq, g.mode, ka, kb, ws.vals_a, ws.vals_b, ws.bag_of, ws.temp, stream);
    }
#ifdef PM_ALTERNATES
    KParams q = p;
    q.bag_begin = 0;
    q.bag_count = p.B;
    q.tiles_per_table = static_cast<int32_t>((p.B + p.bags_per_block - 1) / p.bags_per_block);
    q.xcd_affine = 0;
    const int grid = q.T * q.tiles_per_table;
    const size_t lds = static_cast<size_t>(q.bags_per_block + 2) * sizeof(int64_t);
    // the apply's work-list control words start at zero (the segmented sort's first kernel does this itself)
    hipError_t zrc = hipMemsetAsync(ws.fix_ctl, 0, 4 * sizeof(uint32_t), stream);
    if (zrc != hipSuccess) return zrc;
    const int64_t s0 = p.bag_begin, s1 = p.bag_begin + p.bag_count;
    // per-table segments of a fixed-pooling request, one phase, no weights: bag and table of a lookup follow from its
    // position, so the first radix pass forms the pairs itself from the index array and no key-building kernel runs
    // (33 us and 126 MB of the benchmark step's 190 us sort; PARAM_AMD_SORT_FUSED_KEYS=0 restores it)
    if (g.fused_keys) {
        const RsSource src{p.indices, p.idx64, g.tshift, static_cast<uint32_t>(g.seg_len / p.B)};
        return rs_sort_pairs<K>(ka, kb, ws.vals_a, ws.vals_b, static_cast<size_t>(p.N), nullptr, 0, g.sort_end_bit, ws.temp, stream,
                                static_cast<size_t>(g.seg_len), &src);
    }
    if (g.weighted)
        hipLaunchKernelGGL((build_keys_kernel<K, true>), dim3(grid), dim3(kBlock), lds, stream, q, ka, ws.vals_a,
                           ws.bag_of, g.rbits, g.tshift, g.kbits, g.phase_bags, s0, s1);
    else
        hipLaunchKernelGGL((build_keys_kernel<K, false>), dim3(grid), dim3(kBlock), lds, stream, q, ka, ws.vals_a,
                           ws.bag_of, g.rbits, g.tshift, g.kbits, g.phase_bags, s0, s1);
    hipError_t rc = hipGetLastError();
    if (rc != hipSuccess) return rc;
    size_t tb = ws.temp_bytes;
    // stable LSD radix sort.  rocPRIM leaves the result in keys_b / vals_b, the own sort in the b buffers iff its pass
    // count is odd (plan.in_b).
    if (g.rocprim)
        return rocprim::radix_sort_pairs(ws.temp, tb, ka, kb, ws.vals_a, ws.vals_b, static_cast<size_t>(p.N), 0u,
                                         static_cast<unsigned>(g.sort_end_bit), stream);
    return rs_sort_pairs<K>(ka, kb, ws.vals_a, ws.vals_b, static_cast<size_t>(p.N), nullptr, 0, g.sort_end_bit, ws.temp, stream,
                            g.segmented ? static_cast<size_t>(g.seg_len) : 0);
#else
    return hipErrorInvalidValue;      // (only the segmented sort exists in the product library)
#endif
}

// the workspace is sized for the widest key the request can get (two phases), whatever plan is used later
int ws_key_bytes(const KParams& p, int64_t max_rows) { return (bits_for(max_rows) + 1 + bits_for(p.T) + 1 <= 32) ? 4 : 8; }
int ws_kbits_sort(const KParams& p, int64_t max_rows) { return bits_for(max_rows) + 1 + bits_for(p.T) + 1; }

}  // namespace

// ---- entry points used by capi.hip -----------------------------------------------------------
void set_sort_tuning(int mode) { g_sort_mode.store(mode); }

void set_backward_tuning(int sort_impl, int order, int xcd, int max_phases_) {
    g_sort_impl.store(sort_impl);
    g_sort_order.store(order);
    g_bwd_xcd.store(xcd);
    g_max_phases.store(max_phases_);
}

hipError_t sorted_workspace_bytes(const KParams& p, int64_t max_rows, int max_dim, size_t& bytes) {
    SortWs ws;
    hipError_t rc = ws_layout(nullptr, p.N, p.T, ws_key_bytes(p, max_rows), ws_kbits_sort(p, max_rows), p.psw != nullptr, max_dim, ws);
    bytes = ws.total;
    return rc;
}

hipError_t sort_indices(const KParams& p, int64_t max_rows, int max_dim, int64_t fixed_pooling, int phases, void* workspace,
                        hipStream_t stream, bool defer_ok) {
    SortPlan g = make_plan(p, max_rows, fixed_pooling, phases);
    SortWs ws;
    hipError_t rc = ws_layout(workspace, p.N, p.T, ws_key_bytes(p, max_rows), ws_kbits_sort(p, max_rows), g.weighted, max_dim, ws);
    if (rc != hipSuccess) return rc;
    {
        // Hybrid backward: launched for every unweighted request of the segmented sort large enough to hold an eligible table; which
        // tables take it is decided on the device, from the request alone (a host-side hint fed by the previous sort's verdicts
        // would save ~15 us of empty launches on skewed requests -- and make the bits of rows looked up more than 256 times depend
        // on the history of the workspace: the chunk boundaries of their partial sums follow what else is in the sorted arrays).
        // (The compaction scans one count per tile of the bag-major apply, kCompactMaxTiles = 4096 per table; the apply may tile
        // twice as fine as this call's geometry when its element type differs.)
        // Requests whose lookups do not divide evenly over the bags (ragged bags; per-table pooling factors such as Criteo's) are not
        // offered: there the qualifying tables are a minority in practice (Criteo under uniform indices: 24 % of the lookups) and the
        // launches that find nothing to do cost the 0.4 ms step 3-4 %.  A host-side rule on the request's shape, like everything
        // else here a function of the request alone; enable = 2 (tests) offers every request.
        const int en = hyb_enable_knob();
        const int64_t tb = static_cast<int64_t>(p.T) * p.B;
        const bool even_req = tb > 0 && p.N % tb == 0;
        // ... and only to a sort whose apply follows in the same library call (defer_ok: pm_embbag_bwd_fused*): the hybrid apply reads
        // the request's indices and offsets AGAIN (bag-major kernel, compaction), so a sort issued on its own -- possibly on a side
        // stream, with the caller free to refill the index buffer before the apply -- consumes the request completely, as before round 4.
        const int64_t uniq_tiles = (p.bag_count + kUniqueBags - 1) / kUniqueBags;     // the bag-major apply's tiles per table
        // ... and only to requests whose bag-major launch fills the chip (round 6): the kernel tiles 128 bags, and a request of few
        // tables is a handful of workgroups each pooling 128 bags' lookups in turn -- ONE 10 M-row table, batch 8192, pooling 20: 64
        // workgroups, 254 us against the sorted path's 96; the 40 M-row 100-hot Criteo table on its own 1.29 ms against 0.29.  Same
        // process, hybrid off / on taking turns (tools/r6_few_tables_probe.py, profiles/r06_few_tables_hybrid.jsonl; tables x batch,
        // sorted / hybrid us): 512 tiles 353 / 405 and 326 / 362, 768 tiles 467 / 470, 1024 tiles 653 / 637, 1536 tiles 860 / 830,
        // 2048 tiles 1185 / 1136.  A rule on the request's sizes, like the others here.
        // pm_set_hybrid_min_tiles(0) lifts it (tests drive the hybrid kernels with small requests).
        constexpr int64_t kHybMinTiles = 1024;
        const int min_tiles_knob = g_hyb_min_tiles.load();
        const bool fills = uniq_tiles * p.T >= (min_tiles_knob < 0 ? kHybMinTiles : static_cast<int64_t>(min_tiles_knob));
        if (defer_ok && g.v2 && en > 0 && !g.weighted && p.N >= static_cast<int64_t>(kHybMinCount) && uniq_tiles >= 1 &&
            uniq_tiles <= kCompactMaxTiles && ((even_req && fills) || en >= 2) && seg_sort_hybrid_available())
            g.hyb = en >= 2 ? 2 : 1;
        std::lock_guard<std::mutex> lock(g_plan_mutex);
        // bound the record table: the OLDEST record goes (a clear() would also drop plans of workspaces that are sorted
        // but not yet applied)
        if (g_plans.size() >= 4096 && g_plans.find(workspace) == g_plans.end()) {
            auto oldest = g_plans.begin();
            for (auto it = g_plans.begin(); it != g_plans.end(); ++it)
                if (it->second.stamp < oldest->second.stamp) oldest = it;
            g_plans.erase(oldest);
        }
        SortPlan rec = g;
        rec.stamp = ++g_plan_stamp;
        g_plans[workspace] = rec;
    }
    // the key type follows the PLAN (a one-phase plan of a request whose two-phase key would need 33 bits still sorts
    // 4-byte keys); the buffers were sized for the wider of the two
    return g.key_bytes == 4 ? sort_impl<uint32_t>(p, g, ws, stream) : sort_impl<uint64_t>(p, g, ws, stream);
}

// human-readable form of the plan a sort of this request would use (host-only; tests and sweeps)
std::string sort_plan_describe(const KParams& p, int64_t max_rows, int64_t fixed_pooling, int phases) {
    const SortPlan g = make_plan(p, max_rows, fixed_pooling, phases);
#ifdef PM_ALTERNATES
    const int passes = g.rocprim ? -1 : g.v2 ? seg_sort_passes(g.mode, g.rbits) : rs_num_passes(0, g.sort_end_bit);
#else
    const int passes = g.v2 ? seg_sort_passes(g.mode, g.rbits) : -1;
#endif
    char buf[640];
    if (g.v2) {
        snprintf(buf, sizeof(buf),
                 "sort=seg mode=%d key_bytes=%d rbits=%d hbits=0 kbits=%d sort_bits=%d passes=%d radix_bits=%d lookback=%d local=%d "
                 "segmented=1 segments=device pooling=device phases=1 xcd=%d sliced=%d weighted=%d result_in_b=%d fused_keys=%d",
                 g.mode, g.key_bytes, g.rbits, g.kbits, g.rbits, passes, seg_sort_radix_bits(g.mode, g.rbits),
                 seg_sort_lookback(g.mode, g.rbits, p.N) ? 1 : 0, (g.mode == 1 || g.mode == 2) ? 1 : 0, g.xcd ? 1 : 0, g.sliced ? 1 : 0,
                 g.weighted ? 1 : 0, g.in_b ? 1 : 0, g.fused_keys ? 1 : 0);
        return buf;
    }
    snprintf(buf, sizeof(buf),
             "sort=%s key_bytes=%d rbits=%d hbits=%d kbits=%d sort_bits=%d passes=%d segmented=%d seg_len=%lld phases=%d "
             "apply_seg_tiles=%d xcd=%d sliced=%d weighted=%d result_in_b=%d fused_keys=%d",
             g.rocprim ? "rocprim" : "own", g.key_bytes, g.rbits, g.hbits, g.kbits, g.sort_end_bit, passes, g.segmented ? 1 : 0,
             static_cast<long long>(g.segmented ? g.seg_len : 0), g.H, (g.xcd || g.H > 1) ? g.seg_tiles : 0, g.xcd ? 1 : 0,
             g.sliced ? 1 : 0, g.weighted ? 1 : 0, g.in_b ? 1 : 0, g.fused_keys ? 1 : 0);
    return buf;
}

// 0: ok, 1: no sort was recorded for this workspace / it was for another request, 2: sorted in two bag phases but the
// apply (row-wise Adagrad) needs every row's lookups in ONE run
int bwd_sorted_plan_check(const KParams& p, int64_t max_rows, const void* workspace, bool adagrad) {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    auto it = g_plans.find(workspace);
    if (it == g_plans.end()) return 1;
    const SortPlan& g = it->second;
    if (g.n != p.N || g.T != p.T || g.rbits != bits_for(max_rows) || g.weighted != (p.psw != nullptr) ||
        g.sliced != !(p.bag_begin == 0 && p.bag_count == p.B) || g.indices != p.indices || g.offsets != p.offsets || g.B != p.B ||
        g.bag_begin != p.bag_begin || g.bag_count != p.bag_count)
        return 1;
    if (adagrad && g.H != 1) return 2;
    return 0;
}

// where the last sort on this workspace left its pairs (device pointers into the workspace): tests and tools
int sorted_pairs_info(const KParams& p, int64_t max_rows, int max_dim, const void* workspace, const void** keys, const uint32_t** vals,
                      const uint32_t** d_count, int* key_bytes, int* tshift) {
    SortPlan g;
    {
        std::lock_guard<std::mutex> lock(g_plan_mutex);
        auto it = g_plans.find(workspace);
        if (it == g_plans.end()) return 1;
        g = it->second;
    }
    if (g.hyb && !g.applied) return 2;      // a deferred sort: the pairs exist once the apply has run
    SortWs ws;
    if (ws_layout(const_cast<void*>(workspace), p.N, p.T, ws_key_bytes(p, max_rows), ws_kbits_sort(p, max_rows), g.weighted, max_dim, ws) !=
        hipSuccess)
        return 1;
    *keys = g.in_b ? ws.keys_b : ws.keys_a;
    *vals = g.in_b ? ws.vals_b : ws.vals_a;
    *d_count = g.v2 ? seg_sort_count(ws.temp, static_cast<size_t>(p.N), p.T) : nullptr;
    *key_bytes = g.key_bytes;
    *tshift = g.tshift;
    return 0;
}

hipError_t bwd_sorted_apply(const KParams& p, int64_t max_rows, int dst_dtype, int max_dim, const void* workspace,
                            float* const* momentum, const pm_rowwise_adagrad* opt, hipStream_t stream) {
    SortPlan g;
    {
        std::lock_guard<std::mutex> lock(g_plan_mutex);
        auto it = g_plans.find(workspace);
        if (it == g_plans.end()) return hipErrorInvalidValue;
        g = it->second;
        it->second.applied = true;
    }
    SortWs ws;
    hipError_t rc = ws_layout(const_cast<void*>(workspace), p.N, p.T, ws_key_bytes(p, max_rows), ws_kbits_sort(p, max_rows), g.weighted,
                              max_dim, ws);
    if (rc != hipSuccess) return rc;
    if (p.T > kMaxTablesLds) return hipErrorInvalidValue;
    SortedParams sp;
    sp.recs = ws.recs;
    sp.fix_list = ws.fix_list;
    sp.fix_ctl = ws.fix_ctl;
    sp.partials = ws.partials;
    sp.T = p.T;
    sp.keys = g.in_b ? ws.keys_b : ws.keys_a;
    sp.vals = g.in_b ? ws.vals_b : ws.vals_a;
    sp.bag_of = ws.bag_of;
    sp.dst = const_cast<void* const*>(p.tables);
    sp.dims = p.dims;
    sp.out_offsets = p.out_offsets;
    sp.grad = p.io;
    sp.psw = p.psw;
    sp.out_stride = p.out_stride;
    sp.gblk_shift = p.gblk_shift;
    sp.gblk_extra = p.gblk_extra;
    sp.n = p.N;
    sp.rbits = g.rbits;
    sp.kbits = g.kbits;
    sp.tshift = g.tshift;
    sp.max_dim = max_dim;
    sp.nt_rows = p.nt_loads;
    sp.alpha = p.alpha;
    sp.mom = momentum;
    sp.lr = opt ? opt->lr : 0.0f;
    sp.eps = opt ? opt->eps : 0.0f;
    sp.wd = opt ? opt->weight_decay : 0.0f;
    sp.wd_mode = opt ? opt->weight_decay_mode : PM_WD_NONE;
    sp.sr = (opt && opt->stochastic_rounding && dst_dtype != PM_F32) ? 1 : 0;
    sp.sr_seed = opt ? opt->seed : 0;
    sp.exact_run = kExactRun;
    sp.seg_tiles = (!g.v2 && (g.xcd || g.H > 1)) ? g.seg_tiles : 0;
    sp.H = g.H;
    sp.phase = 0;
    sp.xcd = g.xcd ? (g.v2 ? 2 : 1) : 0;
    sp.d_n = g.v2 ? seg_sort_count(ws.temp, static_cast<size_t>(p.N), p.T) : nullptr;
    sp.tile = g.v2 ? apply_tile(p.N) : kSortTile;      // round 2's plans (segments per table, phases) are laid out for 1024
    sp.unique_wgs_per_cu = 0;      // (the bag-major kernel as a looping grid: measured slower, HISTORY r4; one workgroup per tile)
    if (sp.n == 0) return hipSuccess;
    // (Hybrid sorts leave a few per cent of the lookups to this apply.  Measured on what is left of the uniform benchmark request
    // (225 K pairs): 66 us with 256-position tiles, 60 us with 512 -- the same ~5 G pairs/s as at full size, not a latency chain;
    // a looping grid of 4096 workgroups changed nothing for the small launch and cost the full-size one its XCD-contiguous tile
    // order (Zipf apply +13 %).  So: the request-sized tile, one workgroup per possible tile, as for every other sort.)
    auto sorted_apply = [&](hipStream_t s_) {
        switch (dst_dtype) {
            case PM_F32: return bwd_sorted_launch_f32(sp, g.key_bytes, max_dim, s_);
            case PM_BF16: return bwd_sorted_launch_bf16(sp, g.key_bytes, max_dim, s_);
            default: return bwd_sorted_launch_f16(sp, g.key_bytes, max_dim, s_);
        }
    };
    if (!g.hyb) return sorted_apply(stream);
    // Hybrid: the bag-major kernel applies the rows looked up once and lists the other lookups; then the rest of the sort
    // (compaction of the lists, prep 2, the passes) and the sorted apply of what is left -- a few per cent of the request
    // under uniform indices, everything if no table qualified.
    UniqueArgs ua;
    ua.hyb_tab = seg_sort_hyb_tab(ws.temp, static_cast<size_t>(p.N), p.T);
    ua.bloom = seg_sort_bloom(ws.temp, static_cast<size_t>(p.N), p.T);
    ua.emit_keys = ws.keys_b;
    ua.emit_vals = ws.vals_b;
    ua.key_bytes = g.key_bytes;
    ua.tile_cnt = seg_sort_tile_cnt(ws.temp, static_cast<size_t>(p.N), p.T);
    ua.tile_cnt_stride = seg_sort_tile_cnt_stride(static_cast<size_t>(p.N));
    ua.bloom_wbits = bloom_wbits(hyb_slices(p.N, p.T));
    // The bag-major kernel tiles the request by 128 bags whatever the forward's tiling (32 bags at L = 20; 8 for short-bag
    // requests: 26 624 workgroups for the Criteo request, most of which find a table that did not qualify and leave -- 15 us of
    // dispatch).  Measured at benchmark size, uniform indices, sort + apply ms, fp32 48 tables / bf16 64 tables (round 4, same box
    // per pair): 32 bags 1.622 / --, 64: 1.597 / 1.287, 128: 1.555 / 1.184 (on the box where 64 read 1.617), 256 (8192-entry index
    // tile): 1.676 / 1.232.  A hybrid table's tile of 128 bags is at most 128 x 32 lookups (kHybMaxCount / bags): the 4096-entry
    // LDS index tile; longer tiles take the unstaged path.  Row loads in flight per lane group: 4 (2: same / -1 %, 8: -2.5 / -4 %).
    // The tile is kUniqueBags for every request and element type: the sort's guard (sort_indices), this launch and the compaction
    // of its per-tile lists derive their geometry from that one constant and the request's bag count -- nothing a knob or the
    // environment can make disagree between the two calls.
    KParams q = p;
    q.bags_per_block = kUniqueBags;
    q.tiles_per_table = static_cast<int32_t>((q.bag_count + kUniqueBags - 1) / kUniqueBags);
    q.idx_cap = 4096;
    if (q.tiles_per_table < 1 || q.tiles_per_table > kCompactMaxTiles) return hipErrorInvalidValue;   // (before any table is touched; the sort refused such requests)
    switch (dst_dtype) {
        case PM_F32: rc = bwd_unique_launch_f32(sp, q, ua, max_dim, stream); break;
        case PM_BF16: rc = bwd_unique_launch_bf16(sp, q, ua, max_dim, stream); break;
        default: rc = bwd_unique_launch_f16(sp, q, ua, max_dim, stream); break;
    }
    if (rc != hipSuccess) return rc;
    // What the bag-major kernel listed (round 6): hyb_stage_kernel copies a table's few thousand flagged lookups into the staging area
    // (its sort segment becomes empty) or compacts a longer list for the key sort; hyb_rest_kernel sorts the staged tables in LDS and
    // applies them.  The key sort's chain is launched either way -- what was staged is known on the device only -- and finds no pairs
    // for the staged tables.
    const int rest_mode = hyb_rest_knob();
    const HybTiles tiles{q.bags_per_block, q.tiles_per_table};
    const size_t nN = static_cast<size_t>(p.N);
    if (g.key_bytes == 4) {
        const SegSortRequest rq = seg_request<uint32_t>(p, g, ws);
        rc = seg_sort_stage_leftovers<uint32_t>(rq, reinterpret_cast<const uint32_t*>(ws.keys_b), ws.vals_b, reinterpret_cast<uint32_t*>(ws.keys_a),
                                                ws.vals_a, tiles, rest_mode != 0 ? 1 : 0, ws.temp, stream);
    } else {
        const SegSortRequest rq = seg_request<uint64_t>(p, g, ws);
        rc = seg_sort_stage_leftovers<uint64_t>(rq, reinterpret_cast<const uint64_t*>(ws.keys_b), ws.vals_b, reinterpret_cast<uint64_t*>(ws.keys_a),
                                                ws.vals_a, tiles, rest_mode != 0 ? 1 : 0, ws.temp, stream);
    }
    if (rc != hipSuccess) return rc;
    if (rest_mode != 0) {
        RestArgs ra;
        ra.hyb_tab = ua.hyb_tab;
        ra.T_h = p.T < kHybMaxTables ? p.T : kHybMaxTables;
        ra.stage = seg_sort_rest_stage(ws.temp, nN, p.T);
        ra.rest_n = seg_sort_rest_n(ws.temp, nN, p.T);
        ra.rbits = &seg_sort_desc(ws.temp, nN, p.T)->rbits;
        ra.rbits_stride = static_cast<int>(sizeof(SegDesc) / sizeof(uint32_t));
        ra.parts = rest_parts(ra.T_h, q.tiles_per_table);
        switch (dst_dtype) {
            case PM_F32: rc = bwd_rest_launch_f32(sp, q, ra, max_dim, stream); break;
            case PM_BF16: rc = bwd_rest_launch_bf16(sp, q, ra, max_dim, stream); break;
            default: rc = bwd_rest_launch_f16(sp, q, ra, max_dim, stream); break;
        }
        if (rc != hipSuccess) return rc;
    }
    if (g.key_bytes == 4) {
        const SegSortRequest rq = seg_request<uint32_t>(p, g, ws);
        rc = seg_sort_part_b<uint32_t>(rq, g.mode, reinterpret_cast<uint32_t*>(ws.keys_a), reinterpret_cast<uint32_t*>(ws.keys_b), ws.vals_a,
                                       ws.vals_b, ws.bag_of, ws.temp, stream, true);
    } else {
        const SegSortRequest rq = seg_request<uint64_t>(p, g, ws);
        rc = seg_sort_part_b<uint64_t>(rq, g.mode, reinterpret_cast<uint64_t*>(ws.keys_a), reinterpret_cast<uint64_t*>(ws.keys_b), ws.vals_a,
                                       ws.vals_b, ws.bag_of, ws.temp, stream, true);
    }
    if (rc != hipSuccess) return rc;
    return sorted_apply(stream);

}

void set_hybrid_rest(int mode) { g_hyb_rest.store(mode); }
void set_hybrid_min_tiles(int tiles) { g_hyb_min_tiles.store(tiles); }

// workgroups per table of hyb_rest_kernel: enough to put a workgroup on every CU (T_h x parts ~ 256): a staged table's sorted
// positions are dealt out over them
int rest_parts(int T_h, int) {
    const int p = T_h > 0 ? 256 / T_h : 1;
    return p < 1 ? 1 : (p > 16 ? 16 : p);
}

void set_hybrid_tuning(int enable, uint32_t spin_cap) {
    g_hyb_enable.store(enable);
    g_spin_cap.store(spin_cap);
}

// synchronous: what the last sort on this workspace left on the device
hipError_t sort_status(const KParams& p, int64_t max_rows, int max_dim, const void* workspace, hipStream_t stream, uint32_t out[6]) {
    SortPlan g;
    {
        std::lock_guard<std::mutex> lock(g_plan_mutex);
        auto it = g_plans.find(workspace);
        if (it == g_plans.end()) return hipErrorInvalidValue;
        g = it->second;
    }
    out[0] = out[1] = out[2] = out[3] = out[4] = out[5] = 0;
    if (!g.v2) return hipSuccess;
    SortWs ws;
    hipError_t rc = ws_layout(const_cast<void*>(workspace), p.N, p.T, ws_key_bytes(p, max_rows), ws_kbits_sort(p, max_rows), g.weighted,
                              max_dim, ws);
    if (rc != hipSuccess) return rc;
    if ((rc = hipMemcpyAsync(&out[0], seg_sort_timeouts(ws.temp, static_cast<size_t>(p.N), p.T), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return rc;
    if ((rc = hipMemcpyAsync(&out[1], seg_sort_count(ws.temp, static_cast<size_t>(p.N), p.T), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return rc;
    std::vector<HybTable> tab(static_cast<size_t>(p.T));
    if ((rc = hipMemcpyAsync(tab.data(), seg_sort_hyb_tab(ws.temp, static_cast<size_t>(p.N), p.T), sizeof(HybTable) * tab.size(),
                             hipMemcpyDeviceToHost, stream)) != hipSuccess) return rc;
    if ((rc = hipMemcpyAsync(&out[4], seg_sort_rest_stat(ws.temp, static_cast<size_t>(p.N), p.T), 8, hipMemcpyDeviceToHost, stream)) != hipSuccess) return rc;
    if ((rc = hipStreamSynchronize(stream)) != hipSuccess) return rc;
    for (const HybTable& h : tab) out[2] += h.mode == 1u ? 1u : 0u;
    out[3] = static_cast<uint32_t>(g.hyb);
    return hipSuccess;
}

}  // namespace pm
