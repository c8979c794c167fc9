// Does the HIP runtime on this box clear the AQL barrier bit for hipExtAnyOrderLaunch on gfx950?  Kernel A spins ~200 us; kernel B (launched behind it
// on the SAME stream) stamps its start.  Ordered launch: B starts after A ends.  Any-order: B starts while A runs.
// hipcc --offload-arch=gfx950 -O2 tools/anyorder_test.hip -o /tmp/anyorder_test && /tmp/anyorder_test
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp_kernel(unsigned long long* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = __builtin_amdgcn_s_memrealtime();
}
__global__ void after_kernel(unsigned long long* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[3] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    unsigned long long* d; unsigned long long h[4];
    hipMalloc(&d, 64);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(d, 0, 64, s);
            hipStreamSynchronize(s);
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, d, 20000ull);   // 200 us at 100 MHz
            if (mode == 0) hipLaunchKernelGGL(stamp_kernel, dim3(32), dim3(256), 0, s, d);
            else hipExtLaunchKernelGGL(stamp_kernel, dim3(32), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            hipLaunchKernelGGL(after_kernel, dim3(1), dim3(64), 0, s, d);              // ordered again: must start after BOTH
            hipStreamSynchronize(s);
            hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            printf("%s rep %d: A start 0, A end %.1f us, B start %.1f us, C start %.1f us  (err %d)\n", mode ? "any-order" : "ordered  ", rep, (h[1] - h[0]) / 100.0,
                   ((long long)h[2] - (long long)h[0]) / 100.0, ((long long)h[3] - (long long)h[0]) / 100.0, (int)hipGetLastError());
        }
    }
    return 0;
}
