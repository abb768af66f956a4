// param_amd/csrc/embbag_fwd.hip -- batched EmbeddingBag(sum) forward for CDNA4 / gfx950.
//
// Replaces the kernels the reference reaches at train/compute/pt/pytorch_emb.py:40,61,
// train/comms/pt/dlrm.py:380 (aten::_embedding_bag) and pytorch_dist_backend.py:221,845 /
// split_table_batched_embeddings_ops.py:312 (fbgemm TBE forward).
//
// Memory-bound gather: no MFMA.  Design:
//   * one 256-thread workgroup owns a tile of `bags_per_block` consecutive bags of ONE table;
//     the tile's offsets and its index range are staged into LDS with coalesced loads
//     (indices narrowed to int32), so the dependent row loads never wait on a scattered
//     index load;
//   * a group of G lanes (G*16 B >= one row) owns a bag: each lane keeps a 16-byte column
//     slice of the fp32 accumulator in registers and issues one global_load_dwordx4 per
//     lookup, so a wave64 reads 64/G whole rows per instruction, fully coalesced;
//   * UNROLL independent row loads are issued back to back before the first add
//     (memory-level parallelism: UNROLL KiB in flight per wave at G=32/64; default 2, measured -- the pooling loop is
//     instantiated per index source so that no wait of one path serialises the other);
//   * short-bag requests (per-table pooling such as Criteo's, one-hot tables) run embbag_fwd_flat_kernel below: tiles sized
//     per table on the device, row loads in flight across bag borders;
//   * adds happen in index order per lane => the pooled fp32 result is bit-identical to a
//     sequential sum (and to torch's CPU kernel); no cross-lane reduction is needed;
//   * output rows are written with non-temporal 16-byte stores (never re-read by this kernel);
//   * blockIdx -> (table, tile) can be XCD-affine (common.h) so a table's hot rows stay in
//     one XCD's L2 under Zipf-skewed indices.
#include <type_traits>

#include "common.h"
#include "fwd_elem.h"
#include "rowquant.inc"

namespace pm {
namespace {

using namespace fwd;

// Quantised output burst (p.out_bits = 16 / 8 / 4 / 2): the tile's pooled rows leave LDS as row-wise quantised rows
// (rowquant.hip's formats) -- what the quantised all-to-all sends -- so the fp32 pooled output never reaches HBM.  The
// pooled vector of (bag b, table t) is row b * (out_stride / D) + out_offsets[t] / D of the output: the fp32 layouts
// with D-element rows.  All tables of the request have the same D (pm_embbag_fwd_quantized's contract).
__device__ __forceinline__ float shfl_min(float v, int width) {
    for (int m = width / 2; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, width));
    return v;
}
__device__ __forceinline__ float shfl_max(float v, int width) {
    for (int m = width / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, width));
    return v;
}

template <int BITS>
__device__ __forceinline__ void fused_rows_burst(const float* s_out, int nb, int D, uint8_t* qout, int64_t row0, int64_t rows_per_bag) {
    int gq = 4;
    while (gq * 8 < D) gq *= 2;                               // lanes per row: two 4-column chunks per lane
    const int lane = threadIdx.x % gq;
    const int c0 = lane * 4, c1 = (gq + lane) * 4;
    const bool has0 = c0 < D, has1 = c1 < D;
    const int64_t rb = rq::row_bytes(D, BITS);
    for (int r0 = 0; r0 < nb; r0 += kBlock / gq) {
        const int bg = r0 + threadIdx.x / gq;                 // uniform inside a lane group
        const bool live = bg < nb;
        const float* src = s_out + static_cast<size_t>(live ? bg : 0) * D;
        const f32x4 a = has0 ? *reinterpret_cast<const f32x4*>(src + c0) : f32x4{0, 0, 0, 0};
        const f32x4 b = has1 ? *reinterpret_cast<const f32x4*>(src + c1) : f32x4{0, 0, 0, 0};
        float mn, mx;
        rq::row_min_max(a, b, has0, has1, mn, mx);
        mn = shfl_min(mn, gq);
        mx = shfl_max(mx, gq);
        if (!live) continue;
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        rq::quantize_row_share<BITS>(x, mn, mx, lane == 0, has0, has1, c0, c1, D, qout + (row0 + bg * rows_per_bag) * rb);
    }
}

__device__ __forceinline__ void quantized_burst(const KParams& p, const float* s_out, int nb, int D, int64_t bag0, int64_t out_off) {
    uint8_t* qout = reinterpret_cast<uint8_t*>(p.io);
    const int64_t rows_per_bag = p.out_stride / D;
    const int64_t row0 = bag0 * rows_per_bag + out_off / D;  // row of the tile's first bag
    switch (p.out_bits) {
        case 16: {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const int q = D / 4;                              // 8-byte pieces per row
            for (int i = threadIdx.x; i < nb * q; i += kBlock) {
                const int bg = i / q, c4 = i % q;
                const f32x4 v = *reinterpret_cast<const f32x4*>(s_out + static_cast<size_t>(bg) * D + c4 * 4);
                u32x2 o;
                o.x = rq::half_bits(v.x) | (static_cast<uint32_t>(rq::half_bits(v.y)) << 16);
                o.y = rq::half_bits(v.z) | (static_cast<uint32_t>(rq::half_bits(v.w)) << 16);
                __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(qout + (row0 + bg * rows_per_bag) * 2 * D) + c4);
            }
            break;
        }
        case 8: fused_rows_burst<8>(s_out, nb, D, qout, row0, rows_per_bag); break;
        case 4: fused_rows_burst<4>(s_out, nb, D, qout, row0, rows_per_bag); break;
        default: fused_rows_burst<2>(s_out, nb, D, qout, row0, rows_per_bag); break;
    }
}

template <typename WT, int G, int UNROLL, bool WEIGHTED, bool ORDERED, bool STAGE>
__global__ void __launch_bounds__(kBlock) embbag_fwd_kernel(const KParams p) {
    constexpr int VEC = Elem<WT>::kVec;
    constexpr int NG = kBlock / G;  // bags processed concurrently per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_next;  // next unassigned bag of the tile: lane groups pull bags dynamically (ragged bag sizes)

    int t, tile;
    block_to_tile(p, t, tile);
    if (t >= p.T) return;
    const int64_t bag0 = p.bag_begin + static_cast<int64_t>(tile) * p.bags_per_block;
    const int64_t left_bags = p.bag_begin + p.bag_count - bag0;
    const int nb = left_bags < p.bags_per_block ? static_cast<int>(left_bags) : p.bags_per_block;
    if (threadIdx.x == 0) s_next = NG;  // bags 0 .. NG-1 are taken statically; visible after stage_tile's barrier

    int64_t* s_off;
    int32_t* s_idx;
    float* s_w;
    const bool staged = stage_tile_at<WEIGHTED>(p, t, bag0, nb, smem, s_off, s_idx, s_w);
    const int64_t base = s_off[0];
    // ORDERED (ragged requests; the host picks it when the lookups do not divide evenly over the bags): longest bag first.
    // The lane groups take the tile's bags in descending order of length, so a long bag is never the last thing a
    // workgroup starts (heavy-tailed bag sizes: 60 -> 65 % of the roofline).  Tiles of <= 64 bags (the default is 32) are
    // ordered by ONE wave with a bitonic network of wavefront shuffles on (length, -index) keys; larger tiles rank by
    // counting.  Fixed-size bags skip all of it (it costs them 1 %).  Which group pools which bag, and when, changes no result.
    __shared__ uint16_t s_ord[ORDERED ? 1024 : 1];
    if (ORDERED) {
        if (nb <= kWave) {
            if (threadIdx.x < kWave) {
                const int i = threadIdx.x;
                int64_t len = i < nb ? s_off[i + 1] - s_off[i] : -1;           // padding lanes sort last
                if (len > 0x7fffffffffffLL) len = 0x7fffffffffffLL;
                uint64_t key = (static_cast<uint64_t>(len + 1) << 16) | static_cast<uint64_t>(0xffff - i);
#pragma unroll
                for (int k = 2; k <= kWave; k <<= 1) {
#pragma unroll
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        const uint64_t other = __shfl_xor(key, j, kWave);
                        const bool take_max = ((i & k) == 0) == ((i & j) == 0);   // descending overall
                        key = take_max ? (key > other ? key : other) : (key < other ? key : other);
                    }
                }
                if (i < nb) s_ord[i] = static_cast<uint16_t>(0xffff - (key & 0xffff));
            }
        } else {
            for (int i = threadIdx.x; i < nb; i += kBlock) {
                const int64_t li = s_off[i + 1] - s_off[i];
                int rank = 0;
                for (int j = 0; j < nb; ++j) {
                    const int64_t lj = s_off[j + 1] - s_off[j];
                    rank += (lj > li) || (lj == li && j < i);
                }
                s_ord[rank] = static_cast<uint16_t>(i);
            }
        }
        __syncthreads();
    }
    const int gid = threadIdx.x / G;
    const int lig = threadIdx.x % G;
    const int D = p.dims[t];
    constexpr int ES = 16 / VEC;  // bytes per table element
    const int64_t row_bytes = static_cast<int64_t>(D) * ES;
    const char* W = reinterpret_cast<const char*>(p.tables[t]);
    float* out_t = p.io + p.out_offsets[t];
    const bool nt = p.nt_loads != 0;
    // STAGE: the tile's pooled rows are collected in LDS and leave together when the tile is done -- one burst of
    // bags_per_block rows (16 KB, contiguous in the [T, B, D] layout) instead of one 512-byte row whenever a lane group
    // finishes a bag.  LDS: after the index tile, bags_per_block * D floats.
    float* s_out = reinterpret_cast<float*>(smem + tile_lds_bytes(p.bags_per_block, p.idx_cap, WEIGHTED));
    const bool stage = STAGE && nb <= p.stage_bags;   // (always, with bag-count tiles)

    for (int slot = gid; slot < nb;) {
        const int bg = ORDERED ? s_ord[slot] : slot;
        const int64_t s = s_off[bg];
        const int64_t e = s_off[bg + 1];
        float* orow = out_t + (bag0 + bg) * p.out_stride;

        // The pooling loop is instantiated twice, for indices staged in LDS and for indices read from the request: with the
        // choice made per lookup (`staged ? s_idx[..] : load_index(..)`) both paths meet in front of every row load, and the
        // `s_waitcnt vmcnt(0)` the global-index path needs there made the staged path wait for its PREVIOUS ROW LOAD as well --
        // one row in flight per lane group instead of UNROLL.
        auto pool = [&](auto staged_c) {
            constexpr bool ST = decltype(staged_c)::value;
            for (int c = lig * VEC; c < D; c += G * VEC) {
                const char* Wc = W + static_cast<int64_t>(c) * ES;
                float acc[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;

                int64_t j = s;
                // full batches: UNROLL independent row loads in flight (all indices first, then all rows), then ordered adds
                for (; j + UNROLL <= e; j += UNROLL) {
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
                        for (int k = 0; k < VEC; ++k) acc[k] = WEIGHTED ? fmaf(w[u], f[k], acc[k]) : acc[k] + f[k];
                    }
                }
                // tail (< UNROLL lookups): straight-line loads (positions past the bag read its last lookup again and are
                // dropped), same order of additions
                if (j < e) {
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
                            for (int k = 0; k < VEC; ++k) acc[k] = WEIGHTED ? fmaf(w[u], f[k], acc[k]) : acc[k] + f[k];
                        }
                    }
                }

                if (STAGE && stage) {
                    f32x4* o4 = reinterpret_cast<f32x4*>(s_out + static_cast<size_t>(bg) * D + c);
#pragma unroll
                    for (int k = 0; k < VEC; k += 4) o4[k / 4] = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
                } else {
                    // streaming stores: the pooled row is consumed by another kernel / the all-to-all
                    f32x4* o4 = reinterpret_cast<f32x4*>(orow + c);
#pragma unroll
                    for (int k = 0; k < VEC; k += 4) {
                        f32x4 v = {acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
                        __builtin_nontemporal_store(v, o4 + k / 4);
                    }
                }
            }
        };
        if (staged) pool(std::true_type{}); else pool(std::false_type{});
        // next bag: one LDS atomic per bag, broadcast inside the group.  Which group pools which bag does not
        // change any result (a bag is pooled by exactly one group, in index order), only the balance.
        int nxt = 0;
        if (lig == 0) nxt = atomicAdd(&s_next, 1);
        slot = __shfl(nxt, 0, G);
    }
    if (STAGE && stage) {
        __syncthreads();
        if (p.out_bits == 0) {
            const int q = D / 4;                       // 16-byte pieces per row
            auto piece = [&](int bg, int c4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(s_out + static_cast<size_t>(bg) * D + c4 * 4);
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out_t + (bag0 + bg) * p.out_stride + c4 * 4));
            };
            if ((q & (q - 1)) == 0) {                  // piece i is (i >> lg, i & (q - 1)): no integer division per 16 bytes
                const int lg = 31 - __builtin_clz(static_cast<unsigned>(q));
                for (int i = threadIdx.x; i < nb * q; i += kBlock) piece(i >> lg, i & (q - 1));
            } else {
                for (int i = threadIdx.x; i < nb * q; i += kBlock) piece(i / q, i % q);
            }
        } else {
            quantized_burst(p, s_out, nb, D, bag0, p.out_offsets[t]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Flat-walk forward for requests of SHORT bags whose tables differ in pooling factor (Criteo multi-hot 1 .. 100, average 8):
//   * the tile is sized PER TABLE, on the device: every workgroup reads the two offsets that bound its table's slice, takes the
//     table's average bag length from them and cuts the table into tiles of ~flat_target (256) lookups (NG .. flat_bags bags, a
//     multiple of NG); the launch has the grid of the smallest tile, workgroups past a table's tile count leave after that one
//     round trip.  With 8-bag tiles for every table (what this replaces) the twelve pooling-1 Criteo tables were a quarter of
//     the launch's workgroup-slot time for 5.6 % of its lookups: three round trips of set-up per eight row loads.
//   * a lane group owns a contiguous run of the tile's bags and walks their lookups as ONE sequence, UNROLL row loads in flight
//     ACROSS bag borders (a pooling-1 bag is one load: bag by bag, a group would have one load in flight); a bag's sum leaves
//     when the walk passes its end.  Additions happen in index order inside every bag, from zero: same bits as the other kernel.

//   * (round 6) the LANE GROUP is sized per table too: a request whose tables differ in width (mixed embedding dims, reference
//     dlrm.py:384-385) is launched with G lanes x 16 bytes >= its WIDEST row, and a group of G lanes per bag would leave a D = 16
//     fp32 table 4 busy lanes of 32.  A workgroup therefore cuts its 256 lanes into sub-groups of g = the next power of two >=
//     D_t / VEC lanes (at least kBlock / flat_bags, so that a tile's bags still fit the LDS offsets array; at most G): a D = 16
//     row is one 64-byte load by 4 lanes and a 32-lane slot pools eight bags at once.  The tile stays ~flat_target lookups
//     whatever the width, so the index tile in LDS is what it was.  A bag is pooled by ONE sub-group, additions in index order
//     from zero: the same bits whatever g.  Requests of one width compute g = G.
//   * (round 6) NO EMPTY WORKGROUPS IN THE WAY (flat_compact): the grid used to be T x (the smallest tile's count), and a workgroup
//     past its table's tile count left after one round trip.  The dispatcher is in order: the Criteo request launched 26 624
//     workgroups for ~7 600 tiles, and a working workgroup queued behind hundreds of leavers that each held a slot for a round
//     trip (mixed-dim Criteo: its five wide tables alone 123 us, its 21 narrow ones alone 24 us, together 161 us).  Now every
//     workgroup reads every table's slice bounds (T + 1 offsets: one round trip, as before), keeps the tables' tile sizes and the
//     prefix of their tile counts in LDS, and takes tiles blockIdx, blockIdx + grid, ... of the table-major tile order.  The grid
//     is the host's estimate of the tile count (capi.hip): nearly one tile per workgroup, so the hardware dispatcher still
//     balances the load -- a grid of exactly the resident workgroups, each walking ~5 tiles, fixes which workgroup gets the
//     100-hot table's 3 x heavier tiles and lost 10 % to that imbalance (same process, grids taking turns: Criteo D = 128 uniform
//     203 us old grid / 196 resident set / 174.5-179 at 4096-8192 workgroups; mixed dims 177 / 152-155 / 150: profiles/r06_flat_grid_ab.md).
//     Which workgroup pools which tile changes no result.
// (capi.hip offers the compact launch to requests of up to 1024 tables: 6 bytes of LDS each; larger ones keep the T x tiles grid)
// LDS of the flat-walk kernel: [offsets | index tile (| weights)] [burst buffer: stage_bags x stage_out floats] [compact: prefix, tile sizes]
__host__ __device__ inline size_t flat_tables_offset(const KParams& p, bool weighted) {
    return (tile_lds_bytes(p.bags_per_block, p.idx_cap, weighted) + static_cast<size_t>(p.stage_bags) * p.stage_out * sizeof(float) + 15) / 16 * 16;
}

template <typename WT, int G, int UNROLL, bool WEIGHTED>
__global__ void __launch_bounds__(kBlock) embbag_fwd_flat_kernel(const KParams p) {
    constexpr int VEC = Elem<WT>::kVec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // compact: tiles before table t (s_pref[T] = all tiles) and bags per tile of table t -- behind the tile's arrays and the burst
    // buffer, sized by the launcher for the request's T tables (6 bytes per table: as static arrays for kFlatMaxTables they were
    // 6 KB per workgroup and cost the kernel its seventh workgroup per CU)
    int* const s_pref = reinterpret_cast<int*>(smem + flat_tables_offset(p, WEIGHTED));
    uint16_t* const s_bags = reinterpret_cast<uint16_t*>(s_pref + p.T + 1);

    // lane group and tile of table t: g = the next power of two >= D_t / VEC lanes (at least kBlock / flat_bags, at most G);
    // ~flat_target lookups per tile by the table's average bag, NG .. flat_bags bags, a multiple of NG
    auto geometry = [&](int t, int& D, int& g, int& bags) {
        D = p.dims[t];
        g = kBlock / p.flat_bags;                    // narrowest sub-group the tile geometry allows (host: make_params)
        if (g < 1) g = 1;
        while (g < G && g * VEC < D) g <<= 1;
        const int NG = kBlock / g;
        const int64_t g0 = static_cast<int64_t>(t) * p.B + p.bag_begin;
        const int64_t lo = bag_start_or_end(p, g0), hi = bag_start_or_end(p, g0 + p.bag_count);
        const int64_t avg = p.bag_count > 0 ? (hi - lo + p.bag_count - 1) / p.bag_count : 1;
        bags = static_cast<int>(p.flat_target / (avg > 0 ? avg : 1));
        bags = bags > p.flat_bags ? p.flat_bags : bags;
        bags = bags / NG * NG;
        if (bags < NG) bags = NG;
    };

    auto do_tile = [&](int t, int tile, int D, int g, int bags) {
        const int NG = kBlock / g;                   // bags pooled concurrently by this workgroup
        const int64_t bag0 = p.bag_begin + static_cast<int64_t>(tile) * bags;
        const int64_t left_bags = p.bag_begin + p.bag_count - bag0;
        if (left_bags <= 0) return;
        const int nb = left_bags < bags ? static_cast<int>(left_bags) : bags;

        int64_t* s_off;
        int32_t* s_idx;
        float* s_w;
        const bool staged = stage_tile_at<WEIGHTED>(p, t, bag0, nb, smem, s_off, s_idx, s_w);
        const int64_t base = s_off[0];
        const int gid = threadIdx.x / g;
        const int lig = threadIdx.x % g;
        constexpr int ES = 16 / VEC;
        const int64_t row_bytes = static_cast<int64_t>(D) * ES;
        const char* W = reinterpret_cast<const char*>(p.tables[t]);
        float* out_t = p.io + p.out_offsets[t];
        const bool nt = p.nt_loads != 0;
        float* s_out = reinterpret_cast<float*>(smem + tile_lds_bytes(p.bags_per_block, p.idx_cap, WEIGHTED));
        // small tiles (long bags) leave in one burst, like the other kernel: the buffer holds stage_bags rows of stage_out floats -- a
        // narrow table's tile has more, shorter rows
        const bool stage = p.stage_out > 0 && static_cast<int64_t>(nb) * D <= static_cast<int64_t>(p.stage_bags) * p.stage_out;
        const int per = (nb + NG - 1) / NG;
        const int b_lo = gid * per;
        const int b_hi = b_lo + per < nb ? b_lo + per : nb;

        auto walk = [&](auto staged_c) {
            constexpr bool ST = decltype(staged_c)::value;
            for (int c = lig * VEC; c < D; c += g * VEC) {
                const char* Wc = W + static_cast<int64_t>(c) * ES;
                float acc[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
                auto emit = [&](int bg) {                       // the finished sum of bag bg leaves, the accumulator starts over
                    if (stage) {
                        f32x4* o4 = reinterpret_cast<f32x4*>(s_out + static_cast<size_t>(bg) * D + c);
#pragma unroll
                        for (int k = 0; k < VEC; k += 4) o4[k / 4] = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
                    } else {
                        f32x4* o4 = reinterpret_cast<f32x4*>(out_t + (bag0 + bg) * p.out_stride + c);
#pragma unroll
                        for (int k = 0; k < VEC; k += 4) {
                            f32x4 v = {acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
                            __builtin_nontemporal_store(v, o4 + k / 4);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
                };
                int cur = b_lo;
                int64_t cur_end = s_off[cur + 1];
                const int64_t e = s_off[b_hi];
                for (int64_t j = s_off[b_lo]; j < e; j += UNROLL) {
                    u32x4 raw[UNROLL];
                    float w[UNROLL];
                    int64_t r[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {           // straight-line: positions past the run read its last lookup again
                        const int64_t jj = j + u < e ? j + u : e - 1;
                        r[u] = ST ? static_cast<int64_t>(s_idx[jj - base]) : load_index(p.indices, jj, p.idx64);
                        if (WEIGHTED) w[u] = ST ? s_w[jj - base] : as_global<float>(p.psw)[jj];
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) raw[u] = load16(Wc + row_offset<ST>(r[u], row_bytes), nt);
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        if (j + u < e) {
                            while (j + u >= cur_end) {           // the walk passed the end of bag `cur` (and of any empty bags after it)
                                emit(cur);
                                ++cur;
                                cur_end = s_off[cur + 1];
                            }
                            float f[VEC];
                            Elem<WT>::widen(raw[u], f);
#pragma unroll
                            for (int k = 0; k < VEC; ++k) acc[k] = WEIGHTED ? fmaf(w[u], f[k], acc[k]) : acc[k] + f[k];
                        }
                    }
                }
                for (; cur < b_hi; ++cur) emit(cur);            // the last bag walked, then the empty ones behind it (zeros)
            }
        };
        if (b_lo < b_hi) {
            if (staged) walk(std::true_type{}); else walk(std::false_type{});
        }
        if (stage) {
            __syncthreads();
            const int q = D / 4;
            for (int i = threadIdx.x; i < nb * q; i += kBlock) {
                const int bg = i / q, c4 = i % q;
                const f32x4 v = *reinterpret_cast<const f32x4*>(s_out + static_cast<size_t>(bg) * D + c4 * 4);
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out_t + (bag0 + bg) * p.out_stride + c4 * 4));
            }
        }
    };

    const bool compact = p.flat_compact > 0;
    if (compact) {
        // every table's tile size and count (one thread per table), then the prefix of the counts (one wave)
        for (int t = threadIdx.x; t < p.T; t += kBlock) {
            int D, g, bags;
            geometry(t, D, g, bags);
            s_bags[t] = static_cast<uint16_t>(bags);
            s_pref[t + 1] = static_cast<int>((p.bag_count + bags - 1) / bags);
        }
        __syncthreads();
        if (threadIdx.x < kWave) {
            int running = 0;
            for (int b0 = 0; b0 < p.T; b0 += kWave) {
                const int i = b0 + static_cast<int>(threadIdx.x);
                int v = i < p.T ? s_pref[i + 1] : 0;
#pragma unroll
                for (int d = 1; d < kWave; d <<= 1) {
                    const int o = __shfl_up(v, d, kWave);
                    if (static_cast<int>(threadIdx.x) >= d) v += o;
                }
                if (i < p.T) s_pref[i + 1] = running + v;
                running += __shfl(v, kWave - 1, kWave);
            }
            if (threadIdx.x == 0) s_pref[0] = 0;
        }
        __syncthreads();
    }
    // compact: tiles blockIdx, + grid, ... of the table-major order; otherwise ONE trip -- one workgroup per (table, tile of the
    // smallest size), surplus workgroups leave
    const int total = compact ? s_pref[p.T] : static_cast<int>(blockIdx.x) + 1;
    const int step = compact ? static_cast<int>(gridDim.x) : 1;
    for (int v = blockIdx.x; v < total; v += step) {
        int t, tile, D, g, bags;
        if (compact) {
            int lo = 0, hi = p.T;                    // the table whose tiles [s_pref[t], s_pref[t + 1]) hold v (uniform: LDS broadcasts)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_pref[mid] <= v) lo = mid; else hi = mid;
            }
            t = lo;
            tile = v - s_pref[t];
            D = p.dims[t];
            g = kBlock / p.flat_bags;
            if (g < 1) g = 1;
            while (g < G && g * VEC < D) g <<= 1;
            bags = s_bags[t];
        } else {
            block_to_tile(p, t, tile);
            if (t >= p.T) return;
            geometry(t, D, g, bags);
        }
        do_tile(t, tile, D, g, bags);
        if (compact) __syncthreads();                // the next tile restages the LDS arrays this one's walk and burst read
    }
}

template <typename WT, int G, int UNROLL>
hipError_t launch_w(const KParams& p, hipStream_t stream) {
    const bool weighted = p.psw != nullptr;
    const int grid = p.T * p.tiles_per_table;
    size_t lds = tile_lds_bytes(p.bags_per_block, p.idx_cap, weighted);
    if (p.flat_bags > 0) {   // short-bag requests with per-table pooling (capi.hip decides)
        lds += static_cast<size_t>(p.stage_bags) * p.stage_out * sizeof(float);
        if (p.flat_compact > 0) {
            lds = flat_tables_offset(p, weighted) + static_cast<size_t>(p.T + 1) * 4 + static_cast<size_t>(p.T) * 2;
            // p.flat_compact = the workgroup count capi.hip chose: about as many as the request has tiles (an estimate from its sizes: the
            // tiles' true count is on the device), so nearly every workgroup pools ONE tile and the dispatcher balances the load; where
            // the estimate falls short workgroups walk on (b + grid, ...), where it overshoots the surplus leaves after the prologue --
            // at the END of the dispatch order, behind no one
            const int g2 = p.flat_compact < grid ? p.flat_compact : grid;
            if (weighted) hipLaunchKernelGGL((embbag_fwd_flat_kernel<WT, G, UNROLL, true>), dim3(g2), dim3(kBlock), lds, stream, p);
            else hipLaunchKernelGGL((embbag_fwd_flat_kernel<WT, G, UNROLL, false>), dim3(g2), dim3(kBlock), lds, stream, p);
            return hipGetLastError();
        }
        if (weighted) hipLaunchKernelGGL((embbag_fwd_flat_kernel<WT, G, UNROLL, true>), dim3(grid), dim3(kBlock), lds, stream, p);
        else hipLaunchKernelGGL((embbag_fwd_flat_kernel<WT, G, UNROLL, false>), dim3(grid), dim3(kBlock), lds, stream, p);
        return hipGetLastError();
    }
#define PM_FWD(W_, O_, S_) hipLaunchKernelGGL((embbag_fwd_kernel<WT, G, UNROLL, W_, O_, S_>), dim3(grid), dim3(kBlock), lds, stream, p)
    if (p.stage_out && !p.ordered) {   // fixed-pooling requests (capi.hip decides)
        lds += static_cast<size_t>(p.bags_per_block) * p.stage_out * sizeof(float);   // stage_out = widest row (elements)
        if (weighted) PM_FWD(true, false, true); else PM_FWD(false, false, true);
    } else if (weighted) { if (p.ordered) PM_FWD(true, true, false); else PM_FWD(true, false, false); }
    else { if (p.ordered) PM_FWD(false, true, false); else PM_FWD(false, false, false); }
#undef PM_FWD
    return hipGetLastError();
}

template <typename WT, int G>
hipError_t launch_u(const KParams& p, int unroll, hipStream_t stream) {
    switch (unroll) {
        case 1: return launch_w<WT, G, 1>(p, stream);
        case 2: return launch_w<WT, G, 2>(p, stream);
        case 3: return launch_w<WT, G, 3>(p, stream);
        case 4: return launch_w<WT, G, 4>(p, stream);
        case 6: return launch_w<WT, G, 6>(p, stream);
        default: return launch_w<WT, G, 8>(p, stream);
    }
}

template <typename WT>
hipError_t launch_g(const KParams& p, int max_dim, int unroll, hipStream_t stream) {
    switch (group_lanes(max_dim, Elem<WT>::kVec)) {
        case 8: return launch_u<WT, 8>(p, unroll, stream);
        case 16: return launch_u<WT, 16>(p, unroll, stream);
        case 32: return launch_u<WT, 32>(p, unroll, stream);
        default: return launch_u<WT, 64>(p, unroll, stream);
    }
}

}  // namespace

hipError_t launch_embbag_fwd(const KParams& p, int weight_dtype, int max_dim, int unroll,
                             hipStream_t stream) {
    switch (weight_dtype) {
        case PM_F32: return launch_g<float>(p, max_dim, unroll, stream);
        case PM_BF16: return launch_g<bf16_t>(p, max_dim, unroll, stream);
        default: return launch_g<f16_t>(p, max_dim, unroll, stream);
    }
}

}  // namespace pm
