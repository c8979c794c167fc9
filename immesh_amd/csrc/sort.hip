// Stable device radix sort of (key, value) pairs and the device-wide prefix sum of the stages BEFORE the path (decode, undistortion,
// VoxelGrid, first-scan map build, mesh export).  Plumbing, not a hot kernel: rocPRIM's device-wide LSD radix sort / look-back scan,
// called directly.  The per-scan hot path does not come through here: its scans and sorts are fused into its own kernels (mesh_kernels.hip).  Stability is what map_incremental_grow's "sort by covariance norm, then replay per voxel"
// needs: ties keep ascending scan index, the tie-break the CPU checker uses for std::sort's unspecified order.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "kernels.hpp"
#include "mesh_kernels.hpp"
#include "prof.hpp"

size_t sort_pairs_u64_temp_bytes(int n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const int32_t*)nullptr,
                                    (int32_t*)nullptr, (size_t)n, 0u, 64u, (hipStream_t)0);
    return bytes;
}
size_t sort_pairs_u32_temp_bytes(int n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n,
                                    0u, 32u, (hipStream_t)0);
    return bytes;
}
void sort_pairs_u64(hipStream_t s, void* temp, size_t temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                    const int32_t* vals_in, int32_t* vals_out, int n, int end_bit) {
    KTIMED("radix_sort_pairs_u64", s, (void)rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)end_bit, s));
}
void sort_pairs_u32(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in,
                    int32_t* vals_out, int n, int end_bit) {
    KTIMED("radix_sort_pairs_u32", s, (void)rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)end_bit, s));
}

size_t exclusive_sum_temp_bytes(int n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t)0);
    return bytes;
}
void exclusive_sum_i32(hipStream_t s, void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, int n) {
    KTIMED("exclusive_sum_i32", s, (void)rocprim::exclusive_scan(temp, temp_bytes, in, out, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), s));
}
