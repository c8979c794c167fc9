// Latency micro-benchmarks for the lone-wavefront kernels of this library (tools only): what one dependent step of each kind costs in
// s_memtime ticks (= shader clocks at 2.4 GHz).  One wavefront; every chain is 256 dependent steps, unrolled.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
// the timer reads are tied to the value chain (x) so that the compiler cannot move the work out of the timed region
__device__ __forceinline__ unsigned long long tick(double& x) {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x) :: "memory");
    return t;
}
#define T0 const unsigned long long t0 = tick(x);
#define T1(k) { const unsigned long long t1 = tick(x); if (threadIdx.x == 0) out[k] = t1 - t0; }
__global__ void calib(unsigned long long* out, double* sink, double seed, int lanesel) {
    __shared__ double lds[256];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-9, acc = 0;
    { T0
#pragma unroll
      for (int i = 0; i < N; i++) x = x + 1.0000001;
      T1(0) } acc += x;
    { T0
#pragma unroll
      for (int i = 0; i < N; i++) x = x * 1.0000001;
      T1(1) } acc += x;
    { T0
#pragma unroll
      for (int i = 0; i < N; i++) x = __builtin_fma(x, 1.0000001, 1e-9);
      T1(2) } acc += x;
    { T0
#pragma unroll
      for (int i = 0; i < N; i++) x = 1.0 / (x + 1.5);
      T1(3) } acc += x;
    { T0
#pragma unroll
      for (int i = 0; i < N; i++) x = sqrt(x + 2.0);
      T1(4) } acc += x;
    { // independent adds: 8 chains interleaved (issue rate)
      double y[8];
#pragma unroll
      for (int k = 0; k < 8; k++) y[k] = x + k;
      T0
#pragma unroll
      for (int i = 0; i < N / 8; i++) {
#pragma unroll
          for (int k = 0; k < 8; k++) y[k] = y[k] + 1.0000001;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) x += y[k];
      T1(5) }
    { // LDS round trip: write own slot, read neighbour's (dependent)
      lds[lane] = x; __syncthreads();
      T0
#pragma unroll
      for (int i = 0; i < N; i++) { lds[lane] = x; __syncthreads(); x = lds[(lane + 1) & 63]; __syncthreads(); }
      T1(6) } acc += x;
    { // v_readlane -> VALU dependent chain
      T0
#pragma unroll
      for (int i = 0; i < N; i++) { const int lo = __builtin_amdgcn_readlane(__double2loint(x), lanesel), hi = __builtin_amdgcn_readlane(__double2hiint(x), lanesel); x = x + __hiloint2double(hi, lo); }
      T1(7) } acc += x;
    { // ballot -> scalar popcount -> VALU
      T0
      int c = lane + (int)x;
#pragma unroll
      for (int i = 0; i < N; i++) { const unsigned long long m = __ballot(c & 1); c += __popcll(m) + i; }
      x += c;
      T1(8) }
    { // s_memtime back to back
      unsigned long long s = 0;
      T0
#pragma unroll
      for (int i = 0; i < 64; i++) s += __builtin_readcyclecounter();
      x += (double)(s & 1);
      T1(9) }
    { // float add chain
      T0
      float f = (float)x;
#pragma unroll
      for (int i = 0; i < N; i++) f = f + 1.0000001f;
      x += f;
      T1(10) }
    { // int add chain
      T0
      int q = lane + (int)x;
#pragma unroll
      for (int i = 0; i < N; i++) q = q * 3 + i;
      x += q;
      T1(11) }
    { // uniform taken branch loop
      T0
      for (int i = 0; i < N; i++) { x = x + 1.0000001; asm volatile("" ::: "memory"); }
      T1(12) } acc += x;
    sink[lane] = acc;
}
// dependent random 64-byte-line reads over buffers of growing size (lone wavefront): cache / TLB reach of the gather-bound kernels
__global__ void fill_lines(unsigned long long* buf, size_t nlines) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long k = i * 0x9E3779B97F4A7C15ull; k ^= k >> 29; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 32;
        buf[i * 8] = k;
    }
}
__global__ void chase(const unsigned long long* buf, size_t nlines, int steps, unsigned long long* out) {
    unsigned long long idx = threadIdx.x * 7919ull + 12345ull;
    double x = 0;
    const unsigned long long t0 = tick(x);
    for (int i = 0; i < steps; i++) { const unsigned long long v = buf[(idx % nlines) * 8]; idx = idx * 6364136223846793005ull + v; }
    x += (double)(idx & 1);
    const unsigned long long t1 = tick(x);
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = idx; }
}
int main() {
    {
        unsigned long long* d; unsigned long long h[2];
        hipMalloc(&d, 16);
        const size_t sizes_mb[6] = {4, 64, 1024, 8192, 32768, 98304};
        for (int k = 0; k < 6; k++) {
            unsigned long long* buf = nullptr;
            const size_t bytes = sizes_mb[k] << 20;
            if (hipMalloc(&buf, bytes) != hipSuccess) { printf("hipMalloc %zu MB failed\n", sizes_mb[k]); continue; }
            const size_t nlines = bytes / 64;
            hipLaunchKernelGGL(fill_lines, dim3(4096), dim3(256), 0, 0, buf, nlines);
            for (int lanes = 1; lanes <= 64; lanes *= 64) {
                hipLaunchKernelGGL(chase, dim3(1), dim3(lanes), 0, 0, buf, nlines, 512, d);
                hipDeviceSynchronize();
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                printf("dependent random line reads over %6zu MB, %2d lanes (distinct lines): %8.1f ticks = %6.1f ns per step\n", sizes_mb[k], lanes, (double)h[0] / 512, (double)h[0] / 512 / 2.4);
            }
            hipFree(buf);
        }
        hipFree(d);
    }
    unsigned long long* d; unsigned long long h[16]; double* s;
    hipMalloc(&d, 128); hipMalloc(&s, 512);
    const char* names[13] = {"f64 add", "f64 mul", "f64 fma", "f64 div (1/(x+1.5)) incl. add", "f64 sqrt incl. add", "f64 add, 8 independent chains (per op)", "LDS write+read round trip",
                             "2x v_readlane + f64 add", "ballot + popc + add", "s_memtime (64 back to back, per read)", "f32 add", "i32 mad", "f64 add in a rolled loop (taken branch)"};
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(calib, dim3(1), dim3(64), 0, 0, d, s, 1.0, 5);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
    }
    for (int k = 0; k < 13; k++) printf("%-45s %8.1f ticks per step\n", names[k], (double)h[k] / (k == 9 ? 64 : 256));
    return 0;
}
