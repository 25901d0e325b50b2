// sortpath.hip — large-k and threshold searches.
//
// VectorIndex::search_threshold (vector/index.rs:376-388) asks for
// search(query, len) and then keeps score >= threshold, i.e. the whole
// ordered result list; callers may also pass any k.  Beyond the in-register
// top-k lists (k > TOPK_MAX) the scan writes one 64-bit order key per row and
// the list is produced by a device radix sort (rocPRIM) of those keys.
#include <cstring>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "kernels.hpp"

namespace cx {

size_t sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                         (float *)nullptr, (float *)nullptr, (size_t)n, 0u, 64u, (hipStream_t)0);
    return bytes ? bytes : 16;
}

__global__ __launch_bounds__(256) void emit_sorted_kernel(const uint64_t *keys, const float *sims, uint32_t lim,
                                                          uint32_t *out_rows, float *out_scores, float *out_dists,
                                                          uint32_t *out_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lim) return;
    const uint64_t ki = keys[i];
    if (ki == 0ull) {
        if (i == 0) *out_count = 0;
        return;
    }
    const float dist = distance_of(sims[i]);
    out_rows[i] = key_row(ki);
    out_dists[i] = dist;
    out_scores[i] = score_of(dist);
    if (i + 1 == lim || keys[i + 1] == 0ull) *out_count = i + 1;
}

int launch_sort_select(uint64_t *keys_in, float *sims_in, uint64_t *keys_tmp, float *sims_tmp, uint32_t n,
                       uint32_t k, void *temp, size_t temp_bytes, uint32_t *out_rows, float *out_scores,
                       float *out_dists, uint32_t *out_count, hipStream_t stream) {
    if (n == 0 || k == 0) {
        CX_HIP(hipMemsetAsync(out_count, 0, sizeof(uint32_t), stream));
        return CX_OK;
    }
    CX_HIP(rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_in, keys_tmp, sims_in, sims_tmp, (size_t)n, 0u,
                                          64u, stream));
    const uint32_t lim = k < n ? k : n;
    hipLaunchKernelGGL(emit_sorted_kernel, dim3((lim + 255u) / 256u), dim3(256), 0, stream, keys_tmp, sims_tmp, lim,
                       out_rows, out_scores, out_dists, out_count);
    CX_HIP(hipGetLastError());
    return CX_OK;
}


struct U32ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

size_t scan_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    auto it = rocprim::make_transform_iterator((const uint32_t *)nullptr, U32ToU64());
    (void)rocprim::exclusive_scan(nullptr, bytes, it, (uint64_t *)nullptr, (uint64_t)0, (size_t)n,
                                  rocprim::plus<uint64_t>(), (hipStream_t)0);
    return bytes ? bytes : 16;
}

int launch_exclusive_scan(const uint32_t *in, uint64_t *out, uint32_t n, void *temp, size_t temp_bytes,
                          hipStream_t stream) {
    if (!n) return CX_OK;
    auto it = rocprim::make_transform_iterator(in, U32ToU64());
    CX_HIP(rocprim::exclusive_scan(temp, temp_bytes, it, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), stream));
    return CX_OK;
}

}  // namespace cx
