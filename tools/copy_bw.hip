// copy_bw.hip — the HBM yardstick for the memory-bound stage kernels (gfx950): a plain device-to-device copy of one
// padded 512^3 field (518^3 doubles = 1.11 GB) with 8-byte and 16-byte accesses per lane, one element (pair) per thread
// and grid-stride variants.  Build: hipcc -O3 --offload-arch=gfx950 tools/copy_bw.hip -o tools/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>

template <class T>
__global__ void __launch_bounds__(256) copy_flat(const T* __restrict__ a, T* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <class T, int UNROLL>
__global__ void __launch_bounds__(256) copy_stride(const T* __restrict__ a, T* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        T v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) b[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) b[i] = a[i];
}

template <class F>
double time_ms(F f, int reps = 20) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) f();   // ≈70 ms of load first: the device leaves its idle power state
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const size_t n = 518ull * 518 * 518;         // doubles
    double *a, *b;
    (void)hipMalloc(&a, n * 8); (void)hipMalloc(&b, n * 8);
    (void)hipMemset(a, 1, n * 8);
    const double gb = 2.0 * n * 8 / 1e9;
    auto rep = [&](const char* name, double ms) { printf("%-44s %.3f ms  %.2f TB/s (read + write)\n", name, ms, gb / ms); };
    rep("hipMemcpyDtoD", time_ms([&] { (void)hipMemcpyAsync(b, a, n * 8, hipMemcpyDeviceToDevice, 0); }));
    rep("8 B/lane, one element per thread", time_ms([&] { hipLaunchKernelGGL(copy_flat<double>, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n); }));
    rep("16 B/lane, one pair per thread", time_ms([&] { hipLaunchKernelGGL(copy_flat<double2>, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, (const double2*)a, (double2*)b, n / 2); }));
    for (int blocks : {256 * 8, 256 * 16, 256 * 32}) {
        char nm[96];
        snprintf(nm, sizeof nm, "8 B/lane, grid-stride x4, %d blocks", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_stride<double, 4>), dim3(blocks), dim3(256), 0, 0, a, b, n); }));
        snprintf(nm, sizeof nm, "16 B/lane, grid-stride x4, %d blocks", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_stride<double2, 4>), dim3(blocks), dim3(256), 0, 0, (const double2*)a, (double2*)b, n / 2); }));
    }
    return 0;
}
