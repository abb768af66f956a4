/* examples/c_abi/embbag_demo.c -- the C ABI of libparam_amd.so used from plain C (no Python, no torch).
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi/embbag_demo.c \
 *       -o /tmp/embbag_demo -Lparam_amd -lparam_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/param_amd -Wl,-rpath,/opt/rocm/lib
 *
 * Two tables, a ragged request; forward checked bit for bit against a sequential host sum, deterministic backward
 * (sort + apply) checked against a sequential host scatter-add on rows with one lookup (exact) and 1e-5 elsewhere.
 * Exit code 0 = all checks passed.  This is what a binding in any host language does: device pointers + sizes in,
 * error code out, pm_last_error() for the message.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "param_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define PM_CALL(x) do { int r_ = (x); if (r_ != PM_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, pm_last_error()); return 3; } } while (0)

enum { T = 2, B = 64, D0 = 32, D1 = 64, R0 = 1000, R1 = 333 };

int main(void) {
    printf("%s (ABI %d)\n", pm_build_info(), pm_abi_version());
    const int32_t dims_h[T] = {D0, D1};
    const int64_t rows_h[T] = {R0, R1};
    const int64_t col0_h[T] = {0, D0};
    const int out_stride = D0 + D1;

    /* request: bag b of table t has (b + t) % 5 lookups (some empty); offsets carry T*B + 1 entries */
    int64_t off_h[T * B + 1], n = 0;
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < B; ++b) { off_h[t * B + b] = n; n += (b + t) % 5; }
    off_h[T * B] = n;
    int64_t* idx_h = (int64_t*)malloc((size_t)n * sizeof(int64_t));
    uint64_t s = 88172645463325252ull;
    for (int t = 0; t < T; ++t)
        for (int64_t j = off_h[t * B]; j < off_h[(t + 1) * B]; ++j) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            idx_h[j] = (int64_t)(s % (uint64_t)rows_h[t]);
        }

    void* tab_d[T];
    float* tab_h[T];
    for (int t = 0; t < T; ++t) {
        const size_t bytes = (size_t)rows_h[t] * dims_h[t] * sizeof(float);
        HIP_OK(hipMalloc(&tab_d[t], bytes));
        PM_CALL(pm_fill_random(tab_d[t], rows_h[t] * dims_h[t], PM_F32, 1, 0.0f, 1.0f, 7 + t, NULL));
        tab_h[t] = (float*)malloc(bytes);
        HIP_OK(hipMemcpy(tab_h[t], tab_d[t], bytes, hipMemcpyDeviceToHost));
    }
    void **tabs_dd; int64_t *rows_d, *col0_d, *idx_d, *off_d; int32_t *dims_d, *err_d; float *out_d, *grad_d;
    HIP_OK(hipMalloc((void**)&tabs_dd, sizeof(tab_d)));   HIP_OK(hipMemcpy(tabs_dd, tab_d, sizeof(tab_d), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&rows_d, sizeof(rows_h)));   HIP_OK(hipMemcpy(rows_d, rows_h, sizeof(rows_h), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&dims_d, sizeof(dims_h)));   HIP_OK(hipMemcpy(dims_d, dims_h, sizeof(dims_h), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&col0_d, sizeof(col0_h)));   HIP_OK(hipMemcpy(col0_d, col0_h, sizeof(col0_h), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&idx_d, (size_t)n * 8));      HIP_OK(hipMemcpy(idx_d, idx_h, (size_t)n * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&off_d, sizeof(off_h)));     HIP_OK(hipMemcpy(off_d, off_h, sizeof(off_h), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void**)&out_d, (size_t)B * out_stride * 4));
    HIP_OK(hipMalloc((void**)&grad_d, (size_t)B * out_stride * 4));
    HIP_OK(hipMalloc((void**)&err_d, 4));                  HIP_OK(hipMemset(err_d, 0, 4));

    pm_embbag_batch op;
    memset(&op, 0, sizeof(op));
    op.num_tables = T; op.weight_dtype = PM_F32; op.index_dtype = PM_I64; op.max_dim = D1;
    op.batch = B; op.num_indices = n; op.bag_begin = 0; op.bag_count = B;
    op.tables = (const void* const*)tabs_dd; op.rows = rows_d; op.dims = dims_d; op.out_offsets = col0_d;
    op.out_stride = out_stride; op.indices = idx_d; op.offsets = off_d; op.per_sample_weights = NULL;

    int32_t nerr = -1;
    PM_CALL(pm_embbag_check(&op, err_d, NULL));
    HIP_OK(hipMemcpy(&nerr, err_d, 4, hipMemcpyDeviceToHost));
    if (nerr != 0) { fprintf(stderr, "request check: %d bad entries\n", nerr); return 4; }

    /* forward: bit-exact against the sequential host sum */
    PM_CALL(pm_embbag_fwd(&op, out_d, NULL));
    float* out_h = (float*)malloc((size_t)B * out_stride * 4);
    HIP_OK(hipMemcpy(out_h, out_d, (size_t)B * out_stride * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < B; ++b)
            for (int d = 0; d < dims_h[t]; ++d) {
                float acc = 0.0f;
                for (int64_t j = off_h[t * B + b]; j < off_h[t * B + b + 1]; ++j) acc = acc + tab_h[t][idx_h[j] * dims_h[t] + d];
                if (memcmp(&acc, &out_h[(size_t)b * out_stride + col0_h[t] + d], 4) != 0) { fprintf(stderr, "forward mismatch t=%d b=%d d=%d\n", t, b, d); return 5; }
            }

    /* backward: W -= 0.5 * grad(bag), deterministic (sort + apply) */
    float* grad_h = (float*)malloc((size_t)B * out_stride * 4);
    for (int i = 0; i < B * out_stride; ++i) grad_h[i] = (float)((i * 37) % 101) / 101.0f - 0.5f;
    HIP_OK(hipMemcpy(grad_d, grad_h, (size_t)B * out_stride * 4, hipMemcpyHostToDevice));
    const int64_t ws_bytes = pm_embbag_bwd_sorted_workspace(&op, R0);
    if (ws_bytes < 0) { fprintf(stderr, "workspace: %s\n", pm_last_error()); return 6; }
    void* ws_d;
    HIP_OK(hipMalloc(&ws_d, (size_t)ws_bytes));
    /* step 1: the sort-aside form (the sort is complete on its own); step 2: the fused call of ABI v6 (sort + apply in one call: the
       form in which the library may take the hybrid path).  Both checked bit for bit against sequential host loops. */
    float* ref_h[8];
    for (int t = 0; t < T; ++t) {
        const size_t bytes = (size_t)rows_h[t] * dims_h[t] * sizeof(float);
        ref_h[t] = (float*)malloc(bytes);
        memcpy(ref_h[t], tab_h[t], bytes);
    }
    for (int step = 0; step < 2; ++step) {
        const float alpha = step == 0 ? -0.5f : -0.25f;
        if (step == 0) {
            PM_CALL(pm_embbag_sort_indices(&op, R0, ws_d, ws_bytes, NULL));
            PM_CALL(pm_embbag_bwd_sorted(&op, grad_d, tabs_dd, PM_F32, alpha, R0, ws_d, ws_bytes, NULL));
        } else {
            PM_CALL(pm_embbag_bwd_fused(&op, grad_d, tabs_dd, PM_F32, alpha, R0, ws_d, ws_bytes, NULL));
        }
        HIP_OK(hipDeviceSynchronize());
        for (int t = 0; t < T; ++t) {
            const size_t bytes = (size_t)rows_h[t] * dims_h[t] * sizeof(float);
            float* ref = ref_h[t];
            for (int b = 0; b < B; ++b)
                for (int64_t j = off_h[t * B + b]; j < off_h[t * B + b + 1]; ++j)
                    for (int d = 0; d < dims_h[t]; ++d) {
                        const float stepv = alpha * grad_h[(size_t)b * out_stride + col0_h[t] + d];
                        ref[idx_h[j] * dims_h[t] + d] = ref[idx_h[j] * dims_h[t] + d] + stepv;
                    }
            float* got = (float*)malloc(bytes);
            HIP_OK(hipMemcpy(got, tab_d[t], bytes, hipMemcpyDeviceToHost));
            if (memcmp(got, ref, bytes) != 0) { fprintf(stderr, "backward mismatch in table %d (step %d)\n", t, step); return 7; }
            free(got);
        }
    }

    /* error behaviour: a bad argument returns a negative code and a message, nothing is launched */
    op.max_dim = 7;
    if (pm_embbag_fwd(&op, out_d, NULL) != PM_ERR_UNSUPPORTED || strlen(pm_last_error()) == 0) { fprintf(stderr, "error path\n"); return 8; }
    printf("embbag_demo: forward bit-exact, sorted and fused backward bit-exact (%lld lookups, %d tables)\n", (long long)n, T);
    return 0;
}
