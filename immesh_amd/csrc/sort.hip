// Stable device radix sort of (key, value) pairs.  Plumbing, not a hot kernel: rocPRIM's device-wide LSD radix sort
// (through the hipCUB front end).  Stability is what map_incremental_grow's "sort by covariance norm, then replay per voxel"
// needs: ties keep ascending scan index, the tie-break the CPU checker uses for std::sort's unspecified order.
#include <hipcub/hipcub.hpp>
#include "kernels.hpp"
#include "mesh_kernels.hpp"
#include "prof.hpp"

size_t sort_pairs_u64_temp_bytes(int n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const int32_t*)nullptr,
                                       (int32_t*)nullptr, n, 0, 64, (hipStream_t)0);
    return bytes;
}
size_t sort_pairs_u32_temp_bytes(int n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr, n,
                                       0, 32, (hipStream_t)0);
    return bytes;
}
void sort_pairs_u64(hipStream_t s, void* temp, size_t temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                    const int32_t* vals_in, int32_t* vals_out, int n, int end_bit) {
    KTIMED("radix_sort_pairs_u64", s, (void)hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, s));
}
void sort_pairs_u32(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in,
                    int32_t* vals_out, int n, int end_bit) {
    KTIMED("radix_sort_pairs_u32", s, (void)hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, s));
}

size_t exclusive_sum_temp_bytes(int n) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, n, (hipStream_t)0);
    return bytes;
}
void exclusive_sum_i32(hipStream_t s, void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, int n) {
    KTIMED("exclusive_sum_i32", s, (void)hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, n, s));
}
