// What does s_memtime count on this part?  Ratio Δs_memtime / Δs_memrealtime (100 MHz) for (a) one idle-ish wave spinning on
// s_sleep, (b) every CU busy with dependent fp64 FMAs, (c) a wall-clock-calibrated count of a fixed number of dependent
// v_fma_f64 (4 cycles each when the pipe is free): cycles per FMA in s_memtime ticks and in wall time.
// hipcc --offload-arch=gfx950 -O3 -o memtime_probe memtime_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
__global__ void fma_chain(unsigned long long* out, double* sink, int iters) {
    double x = threadIdx.x * 1e-9, a = 1.0000001, b = 1e-12;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = __builtin_fma(x, a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (x == 123.0) sink[0] = x;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    unsigned long long* d; double* sink;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * 8192); hipMalloc(&sink, 8);
    std::vector<unsigned long long> h(2 * 8192);
    for (int rep = 0; rep < 2; ++rep) {
        spin<<<1, 64>>>(d, 20000); hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 16, hipMemcpyDeviceToHost);
        printf("one sleeping wave:            memtime/realtime = %.3f  (x100 MHz)   over %.1f us\n", (double)h[0] / h[1], h[1] * 0.01);
        // one wave per SIMD chip-wide (1024 waves): dependent chain, 4 cycles per FMA expected
        const int iters = 20000;
        fma_chain<<<1024, 64>>>(d, sink, iters); hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 16 * 1024, hipMemcpyDeviceToHost);
        double sm = 0, sr = 0; for (int i = 0; i < 1024; ++i) { sm += h[2 * i]; sr += h[2 * i + 1]; }
        printf("1 wave/SIMD, dependent FMAs:  memtime/realtime = %.3f   memtime ticks per FMA = %.3f   ns per FMA = %.3f\n", sm / sr, sm / 1024 / (iters * 64.0), sr / 1024 * 10.0 / (iters * 64.0));
        // 8 waves per SIMD chip-wide: pipe saturated, 4 cycles per FMA per wave -> 32 cycles per FMA of one wave
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        fma_chain<<<1024 * 2, 256>>>(d, sink, iters / 4);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("   whole launch: %.3f ms for %.3g wave-FMAs -> %.3f ns per wave-FMA per SIMD\n", ms, 8192.0 * (iters / 4) * 64, ms * 1e6 / (8192.0 * (iters / 4) * 64 / 1024));
        hipMemcpy(h.data(), d, 16 * 2048, hipMemcpyDeviceToHost);
        sm = sr = 0; for (int i = 0; i < 2048; ++i) { sm += h[2 * i]; sr += h[2 * i + 1]; }
        printf("8 waves/SIMD, dependent FMAs: memtime/realtime = %.3f   memtime ticks per FMA = %.3f   ns per FMA = %.3f  (pipe-bound: 8 x 4 cycles)\n", sm / sr, sm / 2048 / (iters / 4 * 64.0), sr / 2048 * 10.0 / (iters / 4 * 64.0));
    }
    return 0;
}
