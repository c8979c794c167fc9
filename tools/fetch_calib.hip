// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for the two access patterns of this library (the guide's
// "x2" correction is stated for wide coalesced reads only).  Three kernels with known byte counts over a 1 GiB table (far larger than the
// 256 MB Infinity Cache, every byte touched at most once per kernel):
//   stream_read   every lane reads one float4, consecutive lanes consecutive addresses          -> N * 16 bytes, full 64-byte sectors used
//   gather16      every lane reads ONE 16-byte record at a random 64-byte-aligned address       -> N * 16 bytes useful, N distinct 64-byte lines touched
//   gather_node   every lane reads 48 bytes spread over a random 384-byte record (3 lines)        -> the registration map's node gather
// usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/fetch_calib     (then a second pass with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t k) { k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 27; k *= 0x94d049bb133111ebull; k ^= k >> 31; return k; }
__global__ void stream_read(const float4* __restrict__ t, size_t n, float* __restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = t[i];
    if (v.x == 123456.f) sink[0] = v.y;
}
__global__ void gather16(const float4* __restrict__ t, size_t n_lines, size_t n, float* __restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // a permutation-like map of i onto the lines (odd multiplier mod 2^k): every line at most once
    const size_t line = (i * 0x9E3779B97F4A7C15ull + 12345) & (n_lines - 1);
    const float4 v = t[line * 4];
    if (v.x == 123456.f) sink[0] = v.y;
}
__global__ void gather_node(const float4* __restrict__ t, size_t n_nodes, size_t n, float* __restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t node = (i * 1000003ull) % n_nodes;   // 1000003 is coprime to the record count: every record at most once
    const float4* p = t + node * 24;           // 384-byte records
    const float4 a = p[0], b = p[8], c = p[16];   // one 16-byte piece in each of three 128-byte-apart lines
    if (a.x + b.x + c.x == 123456.f) sink[0] = a.y;
}
// ---- WRITE_SIZE (round 5): what a SMALL SCATTERED STORE costs.  The mesher's result kernels (mesh_merge_emit: every record goes to its
// global rank -- 12 + 1 + 4 bytes per triangle, 4 + 24 per smoothed vertex --, mesh_chunk_sort, the table clears of mesh_begin_scan) write a few bytes
// per thread at addresses that are not neighbours.  N threads, each storing `bytes` bytes into a 64-byte line of its own; stream_write for comparison.
__global__ void stream_write(float4* __restrict__ t, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
template <int BYTES>
__global__ void scatter_store(unsigned char* __restrict__ t, size_t n_lines, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t line = (i * 0x9E3779B97F4A7C15ull + 12345) & (n_lines - 1);
    unsigned char* p = t + line * 64;
    if (BYTES == 1) p[0] = (unsigned char)i;
    else if (BYTES == 4) *(int*)p = (int)i;
    else if (BYTES == 12) { int* q = (int*)p; q[0] = (int)i; q[1] = (int)i + 1; q[2] = (int)i + 2; }   // (three dword stores of one thread: out_tri[rank * 3 + 0..2])
    else if (BYTES == 24) { double* q = (double*)p; q[0] = (double)i; q[1] = 1.0; q[2] = 2.0; }         // (out_smooth_xyz[rank * 3 + 0..2])
}
// ---- which kernel is charged for a write-back?  small_dirty writes 2 MiB (coalesced: fits the 4 MB L2 of every XCD many times over, so the lines can stay
// dirty in L2 when the kernel ends), noop_after touches nothing.  If WRITE_SIZE of noop_after is not zero, write-backs are charged to the kernel during
// which they happen, not to the one that dirtied the lines.
__global__ void small_dirty(float4* __restrict__ t, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = make_float4(4.f, 3.f, 2.f, (float)i);
}
__global__ void noop_after(float* __restrict__ sink) { if (threadIdx.x == 12345) sink[1] = 1.f; }
int main() {
    const size_t bytes = 1ull << 30, n4 = bytes / 16, n_lines = bytes / 64, n_nodes = bytes / 384;
    float4* t; float* sink;
    hipMalloc(&t, bytes); hipMalloc(&sink, 64);
    hipMemset(t, 0, bytes);
    hipDeviceSynchronize();
    const size_t n_stream = n4, n_g = 1ull << 22, n_node = 1ull << 21;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream_read, dim3((unsigned)((n_stream + 255) / 256)), dim3(256), 0, 0, t, n_stream, sink);
        hipLaunchKernelGGL(gather16, dim3((unsigned)((n_g + 255) / 256)), dim3(256), 0, 0, t, n_lines, n_g, sink);
        hipLaunchKernelGGL(gather_node, dim3((unsigned)((n_node + 255) / 256)), dim3(256), 0, 0, t, n_nodes, n_node, sink);
        hipLaunchKernelGGL(small_dirty, dim3((unsigned)(((1u << 17) + 255) / 256)), dim3(256), 0, 0, t + (size_t)rep * (1u << 17), (size_t)(1u << 17));   // 2 MiB, a fresh region per repetition
        hipLaunchKernelGGL(noop_after, dim3(1), dim3(64), 0, 0, sink);
        hipLaunchKernelGGL(stream_write, dim3((unsigned)((n_g + 255) / 256)), dim3(256), 0, 0, t, n_g);
        hipLaunchKernelGGL(scatter_store<1>, dim3((unsigned)((n_g + 255) / 256)), dim3(256), 0, 0, (unsigned char*)t, n_lines, n_g);
        hipLaunchKernelGGL(scatter_store<4>, dim3((unsigned)((n_g + 255) / 256)), dim3(256), 0, 0, (unsigned char*)t, n_lines, n_g);
        hipLaunchKernelGGL(scatter_store<12>, dim3((unsigned)((n_g + 255) / 256)), dim3(256), 0, 0, (unsigned char*)t, n_lines, n_g);
        hipLaunchKernelGGL(scatter_store<24>, dim3((unsigned)((n_g + 255) / 256)), dim3(256), 0, 0, (unsigned char*)t, n_lines, n_g);
        hipDeviceSynchronize();
    }
    printf("write side, per launch: stream_write %zu bytes (coalesced); scatter_store<1|4|12|24> %zu threads, each 1 / 4 / 12 / 24 bytes into a 64-byte line of its own (useful bytes %zu / %zu / %zu / %zu)\n",
           n_g * 16, n_g, n_g * 1, n_g * 4, n_g * 12, n_g * 24);
    printf("expected per launch: stream_read %zu bytes; gather16 %zu useful bytes in %zu distinct 64-byte lines (%zu bytes of lines); gather_node %zu useful bytes in %zu lines (%zu bytes of lines)\n",
           n_stream * 16, n_g * 16, n_g, n_g * 64, n_node * 48, n_node * 3, n_node * 3 * 64);
    return 0;
}
