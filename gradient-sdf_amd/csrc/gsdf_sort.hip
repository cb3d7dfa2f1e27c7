/*
 * gsdf_sort.hip -- device-side ordering of the marching-cubes output (gsdf_extract_mesh).
 *
 * LayeredMarchingCubesNoColor::computeIsoSurface (mesh/LayeredMarchingCubesNoColor.cpp:354-561) emits its triangles in a
 * z-y-x sweep; k_mesh emits them in hash-table order together with a 64-bit sweep key.  The list is brought into the
 * reference's order by a radix sort of (key, index) pairs and a gather, both on the device, so that the host receives the
 * finished list in one copy (the host-side std::sort of round 2 was a third of the export time of a 10^5-face mesh).
 * rocPRIM's device radix sort is a library primitive (like hipBLASLt for a plain GEMM): sorting is not part of the hot path.
 */
#include "gsdf_kernels.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

hipError_t gsdf_sort_pairs_u64(void* tmp, size_t* tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                               const uint32_t* vals_in, uint32_t* vals_out, size_t n, hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, s);
}

/* the exchange's union of block ids (gsdf_merge.hip): sort all ranks' key arrays, keep one of each */
hipError_t gsdf_sort_keys_u64(void* tmp, size_t* tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out, size_t n,
                              hipStream_t s) {
    return rocprim::radix_sort_keys(tmp, *tmp_bytes, keys_in, keys_out, n, 0, 64, s);
}
hipError_t gsdf_unique_u64(void* tmp, size_t* tmp_bytes, const unsigned long long* sorted_in, unsigned long long* out,
                           unsigned long long* count_out, size_t n, hipStream_t s) {
    return rocprim::unique(tmp, *tmp_bytes, sorted_in, out, count_out, n, rocprim::equal_to<unsigned long long>(), s);
}

__global__ __launch_bounds__(256) void k_iota(uint32_t* idx, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}
void gsdf_launch_iota(hipStream_t s, uint32_t* idx, size_t n) {
    if (n) hipLaunchKernelGGL(k_iota, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, s, idx, n);
}

/* sorted[i] = tris[order[i]], 9 floats each: one lane per float */
__global__ __launch_bounds__(256) void k_gather_tris(const float* __restrict__ tris, const uint32_t* __restrict__ order,
                                                      float* __restrict__ sorted, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 9) return;
    const size_t t = i / 9, k = i - t * 9;
    sorted[i] = tris[(size_t)order[t] * 9 + k];
}
void gsdf_launch_gather_tris(hipStream_t s, const float* tris, const uint32_t* order, float* sorted, size_t n) {
    if (n) hipLaunchKernelGGL(k_gather_tris, dim3((unsigned int)((n * 9 + 255) / 256)), dim3(256), 0, s, tris, order, sorted, n);
}
