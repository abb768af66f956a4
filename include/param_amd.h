/*
 * include/param_amd.h -- C ABI of libparam_amd.so (MI355X / gfx950).
 *
 * The drop-in boundary of the build.  The reference
 * (facebookresearch/param) is 100 % Python and reaches its kernels through
 * torch / fbgemm_gpu Python calls, so "the reference's FFI for this path" is
 * the set of Python call sites below; each entry point names the call site it
 * replaces (paths relative to the reference root).  Plain pointers and sizes
 * only: no torch types, no allocation inside, stream-ordered and asynchronous,
 * re-entrant across distinct streams.  Every function returns 0 on success or
 * a negative PM_ERR_* code; pm_last_error() gives the message of the calling
 * thread's last failure.
 *
 * All device pointers are caller-owned HBM addresses on the current device.
 * pm_stream_t is a hipStream_t passed as an opaque pointer (NULL = the null
 * stream).
 */
#ifndef PARAM_AMD_H
#define PARAM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_ABI_VERSION 7

typedef void* pm_stream_t;

/* element types */
enum {
    PM_F32 = 0,
    PM_BF16 = 1,
    PM_F16 = 2,
    PM_I32 = 10,
    PM_I64 = 11
};

/* error codes */
enum {
    PM_OK = 0,
    PM_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, bad dtype) */
    PM_ERR_UNSUPPORTED = -2, /* valid but not implemented (e.g. dim not a multiple of the vector width) */
    PM_ERR_HIP = -3,         /* a HIP runtime call failed; message carries hipGetErrorString */
    PM_ERR_INDEX = -4        /* pm_embbag_check found an out-of-range index or a bad offset */
};

/*
 * One batched (multi-table) EmbeddingBag request, TBE layout
 * (train/compute/python/workloads/pytorch/split_table_batched_embeddings_ops.py:93-135,
 * 191-208): `indices` holds the per-table index lists concatenated table-major;
 * `offsets` has num_tables*batch entries (a trailing num_tables*batch+1-th entry
 * equal to num_indices is tolerated and never read); bag (t,b) spans
 * [offsets[t*batch+b], offsets[t*batch+b+1]) and the very last bag ends at
 * num_indices (torch include_last_offset=False rule, train/compute/pt/pytorch_emb.py:172-174).
 * A single nn.EmbeddingBag call is the num_tables == 1 case.
 *
 * Output / gradient addressing: the D_t floats of bag (t,b) live at
 *     base + out_offsets[t] + b * out_stride            (element units)
 * so out_offsets[t] = sum_{u<t} D_u, out_stride = sum_t D_t is the TBE layout
 * [batch, sum D] (pytorch_dist_backend.py:224-228) and out_offsets[t] = t*batch*D,
 * out_stride = D is dlrm.py's torch.stack layout [T, batch, D] (dlrm.py:384-387).
 *
 * [bag_begin, bag_begin+bag_count) selects a slice of the batch dimension so a
 * caller can pipeline chunks of the batch against the all-to-all
 * (pytorch_dist_backend.py:214-234 is the reference's per-op pipelining site).
 */
typedef struct pm_embbag_batch {
    int32_t num_tables;           /* T >= 1 */
    int32_t weight_dtype;         /* PM_F32 | PM_BF16 | PM_F16 : element type of every table */
    int32_t index_dtype;          /* PM_I64 | PM_I32 : element type of indices AND offsets */
    int32_t max_dim;              /* max_t dims[t]; every dims[t] must be a multiple of 4 (f32) / 8 (16-bit) */
    int64_t batch;                /* B: bags per table */
    int64_t num_indices;          /* N: total length of `indices` */
    int64_t bag_begin;            /* first bag (per table) to process, 0 <= bag_begin <= batch */
    int64_t bag_count;            /* bags (per table) to process; bag_begin+bag_count <= batch */
    const void* const* tables;    /* device [T]: base pointer of table t, row-major [rows[t], dims[t]] */
    const int64_t* rows;          /* device [T] */
    const int32_t* dims;          /* device [T] */
    const int64_t* out_offsets;   /* device [T], element units (see above) */
    int64_t out_stride;           /* element units (see above) */
    const void* indices;          /* device [N] */
    const void* offsets;          /* device [T*B] or [T*B+1] */
    const float* per_sample_weights; /* device [N] or NULL */
    int64_t fixed_pooling;        /* (ABI v4: only the sort_impl 1 / 2 alternatives read it; the default sort establishes every
                                     table's pooling on the device by reading the offsets.)
                                     L > 0: the caller GUARANTEES every bag has exactly L lookups (offsets[i] == i * L,
                                     num_indices == T * B * L) -- what every benchmark request of the reference looks like;
                                     0: bags may be ragged.  A hint that changes speed only: the sorted backward then knows
                                     where each table's lookups start without reading device memory (per-table sort segments,
                                     XCD-affine and two-phase apply).  pm_embbag_check verifies it; the kernels do not. */
    /* ABI v6: the BLOCKED send layout of a table-wise sharded exchange, [W][T][B_local][D] -- every peer's chunk of the pooled
       output is one contiguous run whose tiles are 16 KB runs (the [B, sum D] layout writes 512-byte pieces 24 KB apart).
       FORWARD: nothing new in the kernels -- the same indices / offsets arrays describe T * W request tables of batch B_local
       (request table t * W + w = the lookups of weight table t for the bags of block w: the arrays are table-major), with
       tables[t * W + w] = table t's pointer and out_offsets[t * W + w] = w * T * B_local * D + t * B_local * D, out_stride = D;
       `table_group` = W tells the launch that W consecutive request tables read ONE weight table, so that they are placed on
       one XCD (speed only).  BACKWARD (pm_embbag_bwd, pm_embbag_bwd_sorted*, pm_embbag_bwd_fused*): the request keeps its T
       weight tables and batch B = W * B_local (a row's lookups must meet in ONE run); the gradient of bag b of table t is read at
       out_offsets[t] + b * out_stride + (b >> grad_block_shift) * grad_block_extra (B_local = 2^grad_block_shift; out_offsets[t] =
       t * B_local * D, out_stride = D, grad_block_extra = (T - 1) * B_local * D).  0 / 0 / 0: no blocking. */
    int32_t table_group;
    int32_t grad_block_shift;
    int64_t grad_block_extra;      /* (needs grad_block_shift >= 1: a block of one bag is the un-blocked layout) */
    /* ABI v7: min_t dims[t] if the caller knows it on the host, 0 = not given (then treated as max_dim).  A hint that changes
       speed only.  The forward sizes its lane groups for max_dim -- G lanes x 16 bytes >= the widest row -- and with one group per
       bag a D = 16 fp32 table of a request whose widest table has D = 128 would keep 4 of every 32 lanes busy.  When
       min_dim says the request is MIXED (its narrowest table needs a smaller power-of-two lane group than its widest) the
       forward runs the flat-walk kernel with the lane group chosen PER TABLE on the device from dims[t] (sub-groups of 4 .. G
       lanes: a D = 16 row is one 64-byte load by 4 lanes, eight bags per 32-lane slot) and tiles sized per table by their lookups
       (~256 lookups, at most 256 bags).
       Results are bit-identical to a max_dim-wide launch (a bag is pooled by one sub-group, additions in index order).  A table
       narrower than min_dim merely wastes lanes.  Reference: mixed embedding dims, train/comms/pt/dlrm.py:384-385 (`mixed_dim`
       -> torch.cat(ly, dim=1)), :506-557. */
    int32_t min_dim;
    int32_t reserved0;             /* 0 */
} pm_embbag_batch;

/* ABI / build identification. */
int pm_abi_version(void);
const char* pm_build_info(void); /* "gfx950 ... " static string */
const char* pm_last_error(void); /* thread-local; "" if none */

/*
 * Forward: out(t,b)[:] = sum_{j in bag (t,b)} psw[j] * table_t[indices[j], :]
 * fp32 accumulation in index order (bit-identical to a sequential fp32 sum;
 * with per_sample_weights each step is one fused multiply-add), fp32 output.
 * Replaces: nn.EmbeddingBag.__call__ at train/compute/pt/pytorch_emb.py:40,61 and
 * train/comms/pt/dlrm.py:380 (T=1 each / one launch for all local tables), and the
 * fbgemm TBE forward at train/comms/pt/pytorch_dist_backend.py:221,845-849 and
 * split_table_batched_embeddings_ops.py:312.
 */
int pm_embbag_fwd(const pm_embbag_batch* op, float* out, pm_stream_t stream);

/*
 * The same lookup for FEW, LONG bags (inference-style requests: a handful of bags of thousands of lookups): one
 * workgroup per bag, the bag's lookups split over the workgroup's lane groups, partial sums combined with wavefront
 * shuffles and through LDS in a fixed order.  Deterministic, but NOT the sequential order: the result agrees with
 * pm_embbag_fwd to fp32 rounding (1e-5 relative to sum |row|), not bit for bit -- hence a separate entry point.
 * Same arguments and layout rules as pm_embbag_fwd; T * bag_count workgroups are launched.
 */
int pm_embbag_fwd_split(const pm_embbag_batch* op, float* out, pm_stream_t stream);

/*
 * ALTERNATES BUILD ONLY (libparam_amd_alt.so, `make alt`, -DPM_ALTERNATES: tests and tools; not in the product library -- the
 * deterministic backward below is the path, and 25 x faster).  Backward scatter-add with hardware atomics:
 *     dst_t[indices[j], :] += alpha * psw[j] * grad(t, bag(j))[:]
 * `grad` is addressed like `out` above.  dst_tables is a device array [T] of
 * destination base pointers with element type dst_dtype and the tables' shapes:
 *   - a dense fp32 gradient buffer with alpha = 1  == torch's dense EmbeddingBag
 *     backward (aten::_embedding_bag_dense_backward; autograd of pytorch_emb.py:179);
 *   - the tables themselves with alpha = -lr       == the fused in-place update the
 *     reference reaches through fbgemm (pytorch_dist_backend.py:854-857,
 *     split_table_batched_embeddings_ops.py:318-324), plain SGD form.
 * Accumulation uses hardware float atomics (order not fixed; duplicates allowed).
 * dst_dtype: PM_F32, or PM_BF16 / PM_F16 (packed 16-bit atomics, one rounding per add).
 */
#ifdef PM_ALTERNATES
int pm_embbag_bwd(const pm_embbag_batch* op, const float* grad, void* const* dst_tables,
                  int32_t dst_dtype, float alpha, pm_stream_t stream);
#endif

/*
 * Deterministic backward without atomics (the default path of the Python modules):
 *   pm_embbag_bwd_sorted_workspace  bytes of caller-owned device scratch for this request
 *                                   (>= 0), or a negative PM_ERR_* code;
 *   pm_embbag_sort_indices          builds one (table,row) key per lookup and sorts them
 *                                   (stable radix sort); depends on indices/offsets only, so
 *                                   it may run on a side stream under the forward pass;
 *   pm_embbag_bwd_sorted            same arithmetic as pm_embbag_bwd, but every destination
 *                                   row is read once, updated in fp32 registers in lookup
 *                                   order and written once: run-to-run reproducible, and
 *                                   bit-identical to a sequential CPU scatter-add for every
 *                                   row looked up at most 256 times in the call (hotter rows
 *                                   are summed as ordered per-chunk partial sums: same value
 *                                   up to fp32 rounding of the association).  Requires
 *                                   pm_embbag_sort_indices on the same workspace and request
 *                                   (stream-ordered before it).  16-bit destinations are
 *                                   widened, accumulated in fp32 and rounded once per row.
 * The apply must be given THE request the sort was issued for, on the same workspace: same indices / offsets POINTERS (the
 * segmented sort leaves its segments and pair counts on the device, keyed to the request's buffers: a copy of the arrays at
 * another address is another request), same batch, bag_begin, bag_count and weights; otherwise PM_ERR_INVALID.
 * max_rows = max_t rows[t] (host value; sizes the sort key).  The request must satisfy
 * num_indices < 2^32 and batch < 2^32.  Replaces the sort + segmented-reduce backward of
 * aten::_embedding_bag_dense_backward and fbgemm's TBE backward (same call sites as
 * pm_embbag_bwd).
 */
int64_t pm_embbag_bwd_sorted_workspace(const pm_embbag_batch* op, int64_t max_rows);
int pm_embbag_sort_indices(const pm_embbag_batch* op, int64_t max_rows, void* workspace,
                           int64_t workspace_bytes, pm_stream_t stream);
/*
 * The same with the number of BAG PHASES the apply may use: phases = 2 lets pm_embbag_bwd_sorted apply the lower and the
 * upper half of the bags in two launches (each re-reads only half a table's gradient rows, which then stay in the L2 of the
 * XCD serving that table) when the request has fixed pooling and its sizes allow; the result is unchanged (a row's
 * lookups are still added in lookup order).  The fused row-wise Adagrad needs every row's lookups in one run: sort with
 * phases = 1 (what pm_embbag_sort_indices does) for pm_embbag_bwd_sorted_adagrad*, which otherwise return PM_ERR_INVALID.
 * The library remembers, per workspace, how the last sort was laid out; the apply call must follow on the same workspace.
 * (`phases` is honoured by sort_impl 2 only: the default segmented sort, sort_impl 0, always lays the request out for a
 * one-launch apply, whatever `phases` says.)
 */
int pm_embbag_sort_indices_ex(const pm_embbag_batch* op, int64_t max_rows, int32_t phases, void* workspace,
                              int64_t workspace_bytes, pm_stream_t stream);
/*
 * Host-only: writes a one-line description of the layout pm_embbag_sort_indices_ex would choose for this request under
 * the current tuning ("sort=own key_bytes=4 ... passes=3 segmented=1 ... xcd=1 ...") into out.  No device access; for
 * tests, sweeps and bug reports.
 */
int pm_embbag_sort_plan(const pm_embbag_batch* op, int64_t max_rows, int32_t phases, char* out, int32_t out_bytes);
int pm_embbag_bwd_sorted(const pm_embbag_batch* op, const float* grad, void* const* dst_tables,
                         int32_t dst_dtype, float alpha, int64_t max_rows, const void* workspace,
                         int64_t workspace_bytes, pm_stream_t stream);
/*
 * ABI v6.  Sort + apply of one request in ONE call (what a backward step that does not hide the sort under other work
 * wants): pm_embbag_sort_indices followed by pm_embbag_bwd_sorted / pm_embbag_bwd_sorted_adagrad_ex on `stream`, same
 * arguments, same results bit for bit.  Because the library itself sequences the two halves, it may DEFER part of the sort
 * into the apply -- the hybrid backward (pm_set_hybrid_tuning): the sort half then only classifies the tables and builds
 * their "looked up twice" maps, and the apply half reads the request's indices and offsets again.  A sort issued on its own
 * (pm_embbag_sort_indices*) never defers: it consumes the request completely, so the caller may reuse the index buffer
 * once the sort has run, as the sort-aside contract above says.
 */
int pm_embbag_bwd_fused(const pm_embbag_batch* op, const float* grad, void* const* dst_tables, int32_t dst_dtype,
                        float alpha, int64_t max_rows, void* workspace, int64_t workspace_bytes, pm_stream_t stream);
/*
 * ABI v4, host-only, for tests and tools: where the last pm_embbag_sort_indices* on this workspace left its pairs.
 * *keys / *vals: device pointers into the workspace (keys of *key_bytes bytes: table << *tshift | row; values: bag within
 * the table, or the lookup position for weighted requests); *d_count: device uint32 holding the number of pairs (batch
 * slices sort only their own lookups), or NULL when it is num_indices.  Valid until the next sort on the workspace.
 */
int pm_embbag_sorted_pairs(const pm_embbag_batch* op, int64_t max_rows, const void* workspace, const void** keys,
                           const uint32_t** vals, const uint32_t** d_count, int32_t* key_bytes, int32_t* tshift);

/*
 * ABI v5.  What the last sort / apply on this workspace left on the device -- SYNCHRONOUS (copies a few words back and waits
 * for `stream`); for tests, tools and bench lines:
 *   lookback_fallbacks look-back walks of the key sort that stopped waiting for a predecessor tile's published digit counts and
 *                      counted that tile's digits themselves.  Harmless (the result is the same): it is how the one-kernel
 *                      passes make progress without assuming anything about the order in which workgroups are dispatched;
 *                      expected 0 on an otherwise idle device
 *   pairs_sorted       lookups that went through the sort (all of them, or the hybrid path's left-overs)
 *   hybrid_tables      tables whose step took the hybrid path (rows looked up once applied bag-major, see pm_set_hybrid_tuning)
 *   hybrid_launched    0: the hybrid kernels were not launched for this sort (off, weighted or tiny request); else the mode they ran in
 *   lds_pairs          (ABI v7) flagged lookups of hybrid tables that were sorted and applied inside LDS by the left-over kernel
 *                      (pm_set_hybrid_rest) and therefore never reached the sort: a hybrid step applies
 *                      num_indices - pairs_sorted - lds_pairs lookups bag-major
 *   lds_tables         (ABI v7) hybrid tables finished that way
 * Replaces nothing of the reference: it reports on the replacement for the sort inside aten::_embedding_bag_dense_backward
 * (call sites as pm_embbag_bwd_sorted).
 */
typedef struct pm_sort_status {
    uint32_t lookback_fallbacks;
    uint32_t pairs_sorted;
    uint32_t hybrid_tables;
    uint32_t hybrid_launched;
    uint32_t lds_pairs;
    uint32_t lds_tables;
} pm_sort_status;
int pm_embbag_sort_status(const pm_embbag_batch* op, int64_t max_rows, const void* workspace, pm_sort_status* out,
                          pm_stream_t stream);

/*
 * Fused backward + exact row-wise Adagrad (the optimizer the reference configures for its TBE ops,
 * train/comms/pt/comms_utils.py:2014 and split_table_batched_embeddings_ops.py:289; algorithm as
 * published by fbgemm_gpu): per touched row r of table t,
 *     G      = sum over the row's lookups of psw[j] * grad(t, bag(j))[:]      (fp32)
 *     m[r]  += (sum_d G[d]^2) / D_t
 *     W[r]  -= lr / (sqrt(m[r]) + eps) * G
 * `momentum` is a device array [T] of fp32 state pointers, one value per row.  Same sort /
 * workspace / determinism contract as pm_embbag_bwd_sorted; max_dim <= 256 (fp32) / 512 (16-bit).
 */
int pm_embbag_bwd_sorted_adagrad(const pm_embbag_batch* op, const float* grad, void* const* tables,
                                 int32_t table_dtype, float* const* momentum, float lr, float eps,
                                 int64_t max_rows, const void* workspace, int64_t workspace_bytes,
                                 pm_stream_t stream);

/*
 * The same with the remaining options the reference's TBE operator passes to the optimizer
 * (train/compute/python/workloads/pytorch/split_table_batched_embeddings_ops.py:289-300: weight_decay,
 * weight_decay_mode, stochastic_rounding=True; algorithm as published by fbgemm_gpu's rowwise_adagrad):
 *     L2        : m[r] += (sum_d (G[d] + wd * W[r,d])^2) / D ;  W[r] = (1 - mult * wd) * W[r] - mult * G
 *     DECOUPLE  : m[r] += (sum_d G[d]^2) / D                 ;  W[r] = (1 - lr * wd)   * W[r] - mult * G
 * with mult = lr / (sqrt(m[r]) + eps).  stochastic_rounding (bf16 / fp16 tables only): the updated row is
 * rounded to the table type with probability proportional to the distance to the two neighbours (unbiased:
 * updates smaller than half an ulp are not lost on average); the random bits are a counter-based function of
 * (seed, table, row, column) -- pass a different seed every step; results are reproducible for a given seed.
 */
enum { PM_WD_NONE = 0, PM_WD_L2 = 1, PM_WD_DECOUPLE = 2 };
typedef struct pm_rowwise_adagrad {
    float lr;
    float eps;
    float weight_decay;
    int32_t weight_decay_mode;     /* PM_WD_* */
    int32_t stochastic_rounding;   /* 0 / 1 */
    int32_t reserved;
    uint64_t seed;
} pm_rowwise_adagrad;
int pm_embbag_bwd_sorted_adagrad_ex(const pm_embbag_batch* op, const float* grad, void* const* tables,
                                    int32_t table_dtype, float* const* momentum, const pm_rowwise_adagrad* opt,
                                    int64_t max_rows, const void* workspace, int64_t workspace_bytes,
                                    pm_stream_t stream);
/* ABI v6: pm_embbag_sort_indices + pm_embbag_bwd_sorted_adagrad_ex in one call (see pm_embbag_bwd_fused: may take the hybrid path). */
int pm_embbag_bwd_fused_adagrad(const pm_embbag_batch* op, const float* grad, void* const* tables, int32_t table_dtype,
                                float* const* momentum, const pm_rowwise_adagrad* opt, int64_t max_rows, void* workspace,
                                int64_t workspace_bytes, pm_stream_t stream);

/*
 * DLRM input redistribution on the device: regroup what the lengths / indices all-to-alls deliver
 * (train/comms/pt/dlrm.py:744-855) -- lengths [world][num_tables][batch] int64 and the indices
 * concatenated block by block in that (rank, table) order -- into the TBE request of the batched
 * kernel: out_indices table-major (within a table rank-major = global sample order) and out_offsets
 * [num_tables*world*batch + 1].  Replaces splitPerTable (dlrm.py:430-504: O(world*tables) Python
 * slicing / torch.cat with host syncs) with two launches and no host sync.
 * scratch: world*num_tables int64 (device).  out_indices must hold as many entries as `indices`.
 */
int pm_dlrm_regroup(const int64_t* lengths, const int64_t* indices, int32_t world_size, int32_t num_tables,
                    int64_t batch, int64_t* out_indices, int64_t* out_offsets, int64_t* scratch,
                    pm_stream_t stream);

/*
 * Validate a request on the device: every index in [0, rows[t]) and offsets
 * monotone within [0, num_indices]; plus what the kernels assume about the
 * per-table device arrays: rows[t] in [1, 2^31), dims[t] a multiple of 4 (fp32) /
 * 8 (16-bit) and <= max_dim, out_offsets[t] a multiple of 4 elements, table base
 * pointers 16-byte aligned.  Writes the number of violations to
 * *d_error_count (device int32, caller-zeroed is NOT required: the call zeroes
 * it first).  torch raises on such inputs (CPU) / device-asserts (GPU); the
 * forward/backward kernels themselves do not check.
 */
int pm_embbag_check(const pm_embbag_batch* op, int32_t* d_error_count, pm_stream_t stream);
/* The same with extra requirements.  PM_CHECK_UNIFORM_DIMS: every table has dims[t] == max_dim and out_offsets[t] is a
 * multiple of max_dim -- what pm_embbag_fwd_quantized assumes. */
enum { PM_CHECK_UNIFORM_DIMS = 1 };
int pm_embbag_check_ex(const pm_embbag_batch* op, int32_t flags, int32_t* d_error_count, pm_stream_t stream);

/*
 * Fill a buffer with counter-based pseudo-random values at HBM write speed:
 *   dist 0: uniform in [lo, hi)    (dlrm tables: U(-1/sqrt(n), 1/sqrt(n)),
 *                                   train/comms/pt/pytorch_dist_backend.py:923-934)
 *   dist 1: normal(mean=lo, std=hi) (nn.EmbeddingBag default N(0,1), pytorch_emb.py:179)
 * Element i depends only on (seed, i): reproducible for any launch geometry.
 * The values are the build's own stream, not torch's generator.
 */
int pm_fill_random(void* dst, int64_t count, int32_t dtype, int32_t dist, float lo, float hi,
                   uint64_t seed, pm_stream_t stream);

/*
 * Tuning knobs of the forward/backward launch (process-wide, mainly for bench
 * sweeps): unroll = rows in flight per lane group (1,2,3,4,6,8; 0 = default = 2),
 * bags_per_block (0 = default), xcd_affine = 1 maps table t to XCD t%8 when
 * num_tables%8==0 (keeps each table's hot rows in one L2), -1 = default.  nt_loads: forward -- any value > 0 = non-temporal
 * row loads; sorted backward -- cache policy of the destination-row accesses: 0 plain, 1 non-temporal, 2 system scope,
 * 3 plain load + agent-scope (sc1) store (the default at -1: the written row leaves the L2, the re-read gradient rows keep
 * it), 4 non-temporal load + sc1 store.  Speed only.
 */
int pm_set_tuning(int32_t unroll, int32_t bags_per_block, int32_t xcd_affine, int32_t nt_loads);

/*
 * stage_out = 1 (default; -1 = default, PARAM_AMD_FWD_STAGE=0 in the environment changes it): the forward collects a
 * tile's pooled rows in LDS and writes them in one burst when the tile is done (fixed-pooling requests whose tile of
 * rows fits 16 KB).  Results are bit-identical either way; it changes how the output write stream mixes with the row
 * reads (uniform indices: 0.69 -> 0.72-0.74 of the HBM peak).
 * flat_grid (ABI v7; -1 = default, PARAM_AMD_FLAT_COMPACT in the environment changes it): launch shape of the flat-walk forward
 * (short-bag and mixed-dim requests).  0 = one workgroup per (table, smallest tile): workgroups past their table's tile count
 * leave (round 3's form); 1 = one set of workgroups sized by the library, each walking the table-major tile order b, b + grid, ...
 * after establishing every table's tile count from the offsets (no workgroup is dispatched for a tile that does not exist);
 * N > 1 = exactly N workgroups (sweeps).
 * flat_target (ABI v7; -1 = default 256, PARAM_AMD_FLAT_TARGET in the environment changes it): lookups per tile the flat-walk forward
 * aims at when it sizes a table's tile from the table's average bag (sweeps).  Results are bit-identical in every setting.
 */
int pm_set_forward_tuning(int32_t stage_out, int32_t flat_grid, int32_t flat_target);

/*
 * Tuning knobs of the sorted backward (process-wide; -1 = default, which the environment can change:
 * PARAM_AMD_SORT=rocprim, PARAM_AMD_SORT_ORDER=row, PARAM_AMD_BWD_XCD=0, PARAM_AMD_BWD_PHASES=2):
 *   sort_impl   0 (default): the segmented sort of round 3 -- per-table segments, per-table pooling factors and (for batch
 *               slices) the pair count are established ON THE DEVICE from the offsets, so ragged / multi-hot / sliced /
 *               weighted requests take the same fast path as fixed-pooling ones and the fixed_pooling field of the request is
 *               not needed (nor trusted).  The product library has this sort and no other: requests of more than 1024
 *               tables are refused (PM_ERR_UNSUPPORTED: split the request), and sort_impl 1 / 2 are PM_ERR_UNSUPPORTED.
 *               ALTERNATES BUILD ONLY (libparam_amd_alt.so, `make alt`): 1: rocPRIM's radix_sort_pairs; 2: round 2's own LSD
 *               sort with host-side plans (pm_radix_sort_pairs; uses fixed_pooling; also serves more than 1024 tables) --
 *               measured alternatives and independent cross-checks for tests/ and tools/.
 *               order / max_phases below apply to 1 and 2 only; xcd_affine to all (sort_impl 0: XCD-contiguous tiles)
 *   order       1 (default): pairs ordered by (table, [bag phase,] row, position); 0: (row, table, position) -- only
 *               the row bits are sorted, the request being table-major already (one radix pass fewer, a slower apply)
 *   xcd_affine  1 (default; needs order 1 and a fixed-pooling request): the apply kernel's tiles of table t run on
 *               XCD t % 8, so one table's gradient rows are fetched into one L2
 *   max_phases  1 (default): every apply is one launch; 2: pm_embbag_sort_indices_ex(phases = 2) lays a
 *               fixed-pooling request out for the two-phase apply (kept as a measured alternative: it pays under
 *               uniform indices only).
 * Placement, pass count and phases change speed only; every setting gives the same result for rows looked up at
 * most 256 times (longer runs: same value up to fp32 association).
 * Settings are read when a request is SORTED; its apply follows what the sort recorded.
 */
int pm_set_backward_tuning(int32_t sort_impl, int32_t order, int32_t xcd_affine, int32_t max_phases);

/*
 * ABI v4.  How the segmented key sort (sort_impl 0) orders a table's pairs (-1 = default; PARAM_AMD_SORT_MODE in the
 * environment changes the default):
 *   0  LSD passes over all row bits: (table, row, position) order; 8 bits per pass, 9 where that saves a pass (17-18 and
 *      25-27 row bits).  ONE kernel per pass: the digit counts of all passes come from one read of the request, and a
 *      tile learns how many pairs precede it from its predecessors' published counts while the pass runs (look-back)
 *   3  the same order from three kernels per pass (histogram, scan, scatter) -- the cross-check of 0, and what requests
 *      of 2^30 lookups or more get
 *   1  ONE global partition pass on the low row digit, then every (table, digit) bucket is sorted by its remaining bits
 *      inside LDS: (table, row & 255, row >> 8, position) order -- equal rows adjacent and in request order, which is all
 *      the apply kernel needs; buckets stay balanced under any skew
 *   2  the same with the partition on the TOP row digit: ascending rows; a skewed head makes its bucket large
 * Speed only: every mode gives the same tables for rows looked up at most 256 times (longer runs: same value up to fp32
 * association, as documented at pm_embbag_bwd_sorted).
 */
int pm_set_sort_tuning(int32_t mode);

/*
 * ABI v5.  The hybrid backward (sort_impl 0, unweighted requests).  A table whose step touches (nearly) every row once -- the
 * uniform-index benchmark request: 98 % of its lookups -- does not need the sort: pm_embbag_sort_indices* classifies every
 * table on the device (pooling factor present, at most 2^18 lookups, lookups <= rows / 8, and a 2048-lookup sample that does
 * not repeat itself), builds a hashed "looked up twice" bitmap (a blocked Bloom filter, no false negatives) per such table in LDS, and hands only the flagged lookups
 * to the sort; pm_embbag_bwd_sorted* apply the others bag-major (one row read-modify-write per lookup, the bag's gradient slice
 * in registers: 2 x D x e instead of 3 x D x e bytes through the CU per lookup) and the flagged ones through the sorted apply.
 * Same results, bit for bit (a row applied bag-major has no other lookup; the flagged lookups keep their order).
 *   enable   -1 default (= 1); 0 off; 1 on for requests whose lookups divide evenly over the bags (every benchmark shape; ragged
 *            and per-table-pooling requests keep the sorted path): the tables are classified at every sort, on the device, from the request alone (so
 *            the same request always takes the same path: nothing is cached or carried from step to step); 2 every structurally
 *            eligible table goes hybrid whatever its indices look like (tests)                         PARAM_AMD_BWD_HYBRID
 *   lookback_spin_cap  polls before a look-back walk of the key sort stops waiting for a predecessor and counts that tile's digits
 *            itself; 0 = default (2^12).  Tests: 1 = the fallback runs wherever a predecessor is a moment late; 0xFFFFFFFF = it
 *            runs for every predecessor of every tile (nothing published is believed).
 * The path is offered by the FUSED entry points only (pm_embbag_bwd_fused*, ABI v6): its sort half runs the classification and
 * the bitmaps, the rest of the sort (of the flagged lookups, and of every lookup of the tables that did not qualify) runs in the
 * apply half, behind the bag-major kernel, which lists the flagged lookups as a by-product -- and reads the request's indices
 * again.  pm_embbag_sort_indices* on their own always sort completely (round 4 deferred there too: a caller that refilled its
 * index buffer between a side-stream sort and the apply would have had stale dup maps applied to new indices).
 * pm_embbag_sorted_pairs after a fused call shows the pairs that went through the sort; between a deferred sort and its apply
 * (which only the library can be) it returns PM_ERR_INVALID.
 */
int pm_set_hybrid_tuning(int32_t enable, int64_t lookback_spin_cap);

/*
 * ABI v7.  What the bag-major kernel of the hybrid backward leaves of a table -- the lookups its dup map flags, ~3 % of a
 * uniform-index request -- used to go through the whole key sort and the sorted apply: nine small dependent launches for a few
 * thousand pairs per table (7 % of the fp32 benchmark step, 15 % of the bf16 one).  mode 1 (the default at -1; PARAM_AMD_HYB_REST=0
 * in the environment changes it): ONE launch finishes them -- every hybrid table with at most 8192 flagged lookups is sorted
 * by row inside LDS (stable: a row's lookups stay in lookup order) and applied run by run, destination row read once and written
 * once, the arithmetic and order of the sorted apply (bit-identical to the sequential oracle for any run length); tables with
 * more left-overs, and everything when mode is 0, take the round-5 route (lists compacted, key sort, sorted apply).  Which
 * tables were finished is decided on the device from the request alone; the sort's launches follow either way and find no pairs
 * for the tables that were.  Read when the apply is issued.  Same results in either mode for rows looked up at most 256 times.
 */
int pm_set_hybrid_rest(int32_t mode);

/*
 * ABI v7.  The hybrid backward is offered only to requests whose bag-major launch fills the chip: the kernel tiles 128 bags, so a
 * request has num_tables * ceil(bag_count / 128) of its workgroups, and with few of them each workgroup pools its 128 bags' lookups
 * one bag after the other while most of the device idles (ONE 10 M-row table, batch 8192, pooling 20: 254 us against the sorted path's
 * 96).  tiles = the fewest such workgroups for which the path is offered: -1 = default (1024: measured break-even at 768 - 1024,
 * profiles/r06_few_tables_hybrid.jsonl), 0 = no lower bound (tests that drive the hybrid kernels with small requests).  A rule on the
 * request's sizes, read when a request is sorted; results are the same either way (rows looked up at most 256 times: bit for bit).
 */
int pm_set_hybrid_min_tiles(int32_t tiles);

/*
 * Forward with a row-wise QUANTISED output: the pooled vector of (bag b, table t) is written as one quantised row
 * (formats below, bitwidth 16 / 8 / 4 / 2) instead of max_dim floats -- what a quantised all-to-all of pooled embeddings
 * sends (--bitwidth of the reference's comms drivers), produced inside the lookup kernel's output burst so the fp32
 * pooled output never reaches HBM.  Same request as pm_embbag_fwd; the output keeps the fp32 layout with one row per
 * pooled vector: row index b * (out_stride / max_dim) + out_offsets[t] / max_dim, rows packed back to back
 * (pm_rows_quantized_bytes(rows, max_dim, bitwidth) bytes in all; pm_rows_dequantize restores them).  Contract: every
 * table has dims[t] == max_dim (a multiple of 8, <= 512) and out_offsets[t], out_stride are multiples of max_dim
 * (pm_embbag_check_ex(PM_CHECK_UNIFORM_DIMS) verifies the device arrays).  Served by the staged forward only: requests it
 * does not take (ragged bags, staging disabled) get PM_ERR_UNSUPPORTED -- run pm_embbag_fwd + pm_rows_quantize then.
 * Values equal pm_rows_quantize(pm_embbag_fwd(...)) bit for bit.
 */
int pm_embbag_fwd_quantized(const pm_embbag_batch* op, void* out, int32_t bitwidth, pm_stream_t stream);

/*
 * Row-wise quantisation of fp32 rows for the quantised all-to-all of pooled embeddings (the reference's --bitwidth
 * {2,4,8,16} / --quant-a2a-embedding-dim flags, train/comms/pt/comms_utils.py:1788-1806; downcast / restore around the
 * exchange, pytorch_dist_backend.py:48-76 and the unpublished all_to_allv_internal at :273).  src: n_rows x dim fp32,
 * row-major, dim a multiple of 8 in [8, 512].  Quantised row layouts (what torch's
 * quantized::embedding_bag_{byte,4bit,2bit}_prepack produce, i.e. fbgemm's fused row-wise formats):
 *   16: dim x fp16 (round to nearest even)                      2 * dim bytes
 *    8: dim x u8, fp32 scale, fp32 bias                         dim + 8 bytes
 *  4/2: dim * bits / 8 x u8 (low bits first), fp16 scale, bias  dim * bits / 8 + 4 bytes
 * pm_rows_dequantize restores x = fma(code, scale, bias).  Rows are packed back to back; the fp32 buffer is 16-byte aligned,
 * the quantised one aligned to the format's word (16 / 8 / 4 / 2 bytes for 16 / 8 / 4 / 2 bits: any whole number of rows into
 * an aligned buffer qualifies, so per-peer chunks of an exchange buffer can be produced in place).
 * pm_rows_quantized_bytes returns n_rows * row bytes (or a negative PM_ERR_* code).
 */
int64_t pm_rows_quantized_bytes(int64_t n_rows, int32_t dim, int32_t bitwidth);
int pm_rows_quantize(const float* src, int64_t n_rows, int32_t dim, int32_t bitwidth, void* dst, pm_stream_t stream);
int pm_rows_dequantize(const void* src, int64_t n_rows, int32_t dim, int32_t bitwidth, float* dst, pm_stream_t stream);

/*
 * ALTERNATES BUILD ONLY (round 2's sort: the product library sorts per-table segments with csrc/seg_sort.hip).
 * Stable LSD radix sort of (key, uint32 value) pairs by key bits [begin_bit, end_bit) -- the sort the sorted
 * backward ran on its (table,row) keys in round 2 (replaces the sort inside aten::_embedding_bag_dense_backward /
 * fbgemm's TBE backward at the call sites of pm_embbag_bwd_sorted).  key_bytes: 4 or 8.  The pairs start in
 * (keys_a, vals_a); 8-bit passes alternate between the a and b buffers and *result_in_b says where the sorted
 * pairs ended up.  n_max <= 2^32 - 1 elements; if d_count is not NULL the number of elements is read from that
 * device uint32 when the kernels run (clamped to n_max), so a producer kernel can decide it without a host
 * round trip.  segment_len > 0 (a multiple of 4096 dividing n_max, d_count NULL): the array is a sequence of segments of
 * that many pairs, each sorted on its own.  scratch: pm_radix_sort_scratch_bytes(n_max) bytes of device memory.
 */
#ifdef PM_ALTERNATES
int64_t pm_radix_sort_scratch_bytes(int64_t n_max);
int pm_radix_sort_pairs(void* keys_a, void* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t n_max,
                        const uint32_t* d_count, int32_t key_bytes, int32_t begin_bit, int32_t end_bit,
                        int64_t segment_len, void* scratch, int64_t scratch_bytes, int32_t* result_in_b,
                        pm_stream_t stream);
#endif


#ifdef __cplusplus
}
#endif
#endif /* PARAM_AMD_H */
