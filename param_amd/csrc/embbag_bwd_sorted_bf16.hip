// param_amd/csrc/embbag_bwd_sorted_bf16.hip -- the sorted backward's apply kernels for bf16 destination tables (one translation
// unit per destination dtype: see bwd_sorted_apply.h).
#include "bwd_sorted_apply.h"
#include "pm_experiments.h"

namespace pm {
namespace {
#include "bwd_sorted_apply_impl.inc"
}  // namespace

hipError_t bwd_sorted_launch_bf16(const SortedParams& sp, int key_bytes, int max_dim, hipStream_t stream) {
    return key_bytes == 4 ? launch_apply_g<SDstBF16, uint32_t>(sp, max_dim, stream) : launch_apply_g<SDstBF16, uint64_t>(sp, max_dim, stream);
}

hipError_t bwd_unique_launch_bf16(const SortedParams& sp, const KParams& kp, const UniqueArgs& ua, int max_dim, hipStream_t stream) {
    return launch_unique_g<SDstBF16>(sp, kp, ua, max_dim, stream);
}

hipError_t bwd_rest_launch_bf16(const SortedParams& sp, const KParams& kp, const RestArgs& ra, int max_dim, hipStream_t stream) {
    return launch_rest_g<SDstBF16>(sp, kp, ra, max_dim, stream);
}

}  // namespace pm

PM_DEFINE_TRACE_READER(pm_experiment_trace_apply_bf16)      // experiment builds only (pm_experiments.h); nothing in the product
