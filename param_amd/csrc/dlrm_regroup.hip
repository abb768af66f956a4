// param_amd/csrc/dlrm_regroup.hip -- DLRM input redistribution, device side.
//
// After the lengths / indices all-to-alls of SparseDataDist (reference train/comms/pt/dlrm.py:744-855) a rank
// holds, for its F local tables and the GLOBAL batch (W ranks x B samples):
//     lengths  [W][F][B]      (rank-major, then table, then sample)
//     indices  concatenated block by block in the same (rank, table) order
// and needs them per table for the lookup.  The reference regroups with O(W*F) Python slicing / torch.cat
// and several host syncs per batch (splitPerTable, dlrm.py:430-504).  Here: two launches, no host sync.
//     regroup_block_sums   one workgroup per (rank, table) block: number of indices in the block
//     regroup_scatter      one workgroup per block (x copy slices): source start (prefix in (r,f) order),
//                          destination start (prefix in (f,r) order), exclusive scan of the block's B lengths
//                          into the TBE offsets array [F*W*B + 1], coalesced copy of the block's indices
// Output = the TBE request of the batched kernel: indices table-major (within a table rank-major = global
// sample order), offsets running on across tables.
#include "common.h"

namespace pm {
namespace {

constexpr int kMaxBlocks = 4096;  // W * F

__global__ void __launch_bounds__(kBlock) regroup_block_sums(const int64_t* lengths, int64_t B, int64_t* block_sizes) {
    __shared__ int64_t s_part[kBlock / kWave];
    const int64_t blk = blockIdx.x;
    const int64_t* l = lengths + blk * B;
    int64_t acc = 0;
    for (int64_t i = threadIdx.x; i < B; i += kBlock) acc += l[i];
    for (int off = kWave / 2; off >= 1; off >>= 1) acc += __shfl_down(acc, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) s_part[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < kBlock / kWave; ++w) t += s_part[w];
        block_sizes[blk] = t;
    }
}

// inclusive scan of one value per thread across the workgroup (wave shuffles + LDS for the 4 wave totals)
__device__ __forceinline__ int64_t block_inclusive_scan(int64_t v, int64_t* s_wave, int64_t& total) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int off = 1; off < kWave; off <<= 1) {
        const int64_t n = __shfl_up(v, off, kWave);
        if (lane >= off) v += n;
    }
    if (lane == kWave - 1) s_wave[wave] = v;
    __syncthreads();
    int64_t base = 0;
    total = 0;
    for (int w = 0; w < kBlock / kWave; ++w) {
        if (w < wave) base += s_wave[w];
        total += s_wave[w];
    }
    __syncthreads();
    return v + base;
}

__global__ void __launch_bounds__(kBlock) regroup_scatter(const int64_t* lengths, const int64_t* indices,
                                                          const int64_t* block_sizes, int W, int F, int64_t B,
                                                          int64_t* out_indices, int64_t* out_offsets) {
    __shared__ int64_t s_red[2][kBlock / kWave];
    __shared__ int64_t s_wave[kBlock / kWave];
    const int blk = blockIdx.x;  // = r * F + f
    const int r = blk / F, f = blk % F;
    // source start: blocks before (r,f) in rank-major order; destination start: blocks before (f,r) in table-major order
    int64_t src = 0, dst = 0;
    for (int i = threadIdx.x; i < W * F; i += kBlock) {
        const int ri = i / F, fi = i % F;
        const int64_t sz = block_sizes[i];
        if (i < blk) src += sz;
        if (fi < f || (fi == f && ri < r)) dst += sz;
    }
    for (int off = kWave / 2; off >= 1; off >>= 1) {
        src += __shfl_down(src, off, kWave);
        dst += __shfl_down(dst, off, kWave);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_red[0][threadIdx.x / kWave] = src;
        s_red[1][threadIdx.x / kWave] = dst;
    }
    __syncthreads();
    src = 0;
    dst = 0;
    for (int w = 0; w < kBlock / kWave; ++w) {
        src += s_red[0][w];
        dst += s_red[1][w];
    }
    const int64_t size = block_sizes[blk];

    if (blockIdx.y == 0) {
        // offsets of the block's B bags: exclusive scan of its lengths, shifted by the destination start
        const int64_t* l = lengths + static_cast<int64_t>(blk) * B;
        int64_t* o = out_offsets + (static_cast<int64_t>(f) * W + r) * B;
        int64_t carry = 0;
        for (int64_t i0 = 0; i0 < B; i0 += kBlock) {
            const int64_t i = i0 + threadIdx.x;
            const int64_t v = i < B ? l[i] : 0;
            int64_t total;
            const int64_t inc = block_inclusive_scan(v, s_wave, total);
            if (i < B) o[i] = dst + carry + inc - v;
            carry += total;
        }
        if (blk == W * F - 1 && threadIdx.x == 0) {
            // trailing entry = total number of indices (the (f=F-1, r=W-1) block ends the table-major order)
            int64_t all = 0;
            for (int i = 0; i < W * F; ++i) all += block_sizes[i];
            out_offsets[static_cast<int64_t>(F) * W * B] = all;
        }
    }
    // coalesced copy of the block's indices, split over gridDim.y slices
    const int64_t per = (size + gridDim.y - 1) / gridDim.y;
    const int64_t lo = per * blockIdx.y, hi = (lo + per < size) ? lo + per : size;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) out_indices[dst + i] = indices[src + i];
}

}  // namespace

hipError_t launch_dlrm_regroup(const int64_t* lengths, const int64_t* indices, int W, int F, int64_t B,
                               int64_t* out_indices, int64_t* out_offsets, int64_t* scratch, hipStream_t stream) {
    if (W * F > kMaxBlocks) return hipErrorInvalidValue;
    hipLaunchKernelGGL(regroup_block_sums, dim3(W * F), dim3(kBlock), 0, stream, lengths, B, scratch);
    hipError_t rc = hipGetLastError();
    if (rc != hipSuccess) return rc;
    const int slices = W * F >= 512 ? 1 : (W * F >= 64 ? 4 : 16);  // enough workgroups for the copy
    hipLaunchKernelGGL(regroup_scatter, dim3(W * F, slices), dim3(kBlock), 0, stream, lengths, indices, scratch, W, F, B,
                       out_indices, out_offsets);
    return hipGetLastError();
}

}  // namespace pm
