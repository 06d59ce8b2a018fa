// ubench2.hip — issue cost AND sustained clock of the instruction kinds the stage kernel is made of (gfx950).
// For each instruction kind: every CU runs W waves per SIMD of a long stream of that instruction (8 independent
// accumulators), repeated back to back for ~0.4 s so that DVFS settles; the last launch is stamped with
// s_memtime (shader cycles) and s_memrealtime (100 MHz) per wave:
//     cycles per instruction per SIMD = Δmemtime / (W · ITER · 8)
//     clock under this load           = Δmemtime / Δmemrealtime × 100 MHz       (MI355X_MICROARCH.md, DVFS item 6)
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench2.hip -o tools/ubench2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define ITER 4096

#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OPS_D : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c)
#define OPS_I : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib), "v"(ib2)

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* cyc, unsigned long long* rt, double b, double c) {
    __shared__ double lds[1024];
    double x0 = threadIdx.x * 1e-3 + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    const int ib = (int)b, ib2 = (int)(c * 1e9);
    lds[threadIdx.x] = x0; lds[threadIdx.x + 256] = x1; lds[threadIdx.x + 512] = x2; lds[threadIdx.x + 768] = x3;
    __syncthreads();
    const unsigned la = threadIdx.x * 8;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITER; ++i) {
#define S0(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
        if constexpr (OP == 0) asm volatile(R8(S0) OPS_D);
#define S1(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
        if constexpr (OP == 1) asm volatile(R8(S1) OPS_D);
#define S2(n) "v_add_f64 %" #n ", %" #n ", %8\n"
        if constexpr (OP == 2) asm volatile(R8(S2) OPS_D);
#define S3(n) "v_max_f64 %" #n ", %" #n ", %8\n"
        if constexpr (OP == 3) asm volatile(R8(S3) OPS_D);
#define S4(n) "v_mov_b64 %" #n ", %8\n"
        if constexpr (OP == 4) asm volatile(R8(S4) OPS_D);
#define S5(n) "v_mov_b32 %" #n ", %8\n"
        if constexpr (OP == 5) asm volatile(R8(S5) OPS_I);
#define S6(n) "v_mov_b32_dpp %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        if constexpr (OP == 6) asm volatile(R8(S6) OPS_I);
#define S7(n) "v_mov_b32_dpp %" #n ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
        if constexpr (OP == 7) asm volatile(R8(S7) OPS_I);
#define S8(n) "v_rcp_f64 %" #n ", %" #n "\n"
        if constexpr (OP == 8) asm volatile(R8(S8) OPS_D);
#define S9(n) "v_rsq_f64 %" #n ", %" #n "\n"
        if constexpr (OP == 9) asm volatile(R8(S9) OPS_D);
#define S10(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
        if constexpr (OP == 10) asm volatile(R8(S10) OPS_D);
#define S11(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
        if constexpr (OP == 11) asm volatile(R8(S11) OPS_I);
#define S12(n) "v_add_u32 %" #n ", %" #n ", %8\n"
        if constexpr (OP == 12) asm volatile(R8(S12) OPS_I);
        // LDS reads only (8 ds_read_b64 per iteration, drained once per iteration)
        if constexpr (OP == 13)
            asm volatile("ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:2048\nds_read_b64 %2, %8 offset:4096\nds_read_b64 %3, %8 offset:6144\n"
                         "ds_read_b64 %4, %8\nds_read_b64 %5, %8 offset:2048\nds_read_b64 %6, %8 offset:4096\nds_read_b64 %7, %8 offset:6144\n"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(la));
        // 8 fma_f64 + 4 ds_read_b64 per iteration: does the LDS read stream cost VALU issue?
        if constexpr (OP == 14) {
            double y0, y1, y2, y3;
            asm volatile("ds_read_b64 %0, %4\nds_read_b64 %1, %4 offset:2048\nds_read_b64 %2, %4 offset:4096\nds_read_b64 %3, %4 offset:6144\n"
                         : "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3) : "v"(la));
            asm volatile(R8(S0) OPS_D);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(y0), "v"(y1), "v"(y2), "v"(y3));
        }
        // 8 fma_f64 + 8 ds_read_b64
        if constexpr (OP == 15) {
            double y0, y1, y2, y3, y4, y5, y6, y7;
            asm volatile("ds_read_b64 %0, %8\nds_read_b64 %1, %8 offset:2048\nds_read_b64 %2, %8 offset:4096\nds_read_b64 %3, %8 offset:6144\n"
                         "ds_read_b64 %4, %8 offset:8\nds_read_b64 %5, %8 offset:2056\nds_read_b64 %6, %8 offset:4104\nds_read_b64 %7, %8 offset:6152\n"
                         : "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3), "=v"(y4), "=v"(y5), "=v"(y6), "=v"(y7) : "v"(la));
            asm volatile(R8(S0) OPS_D);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(y4), "v"(y5), "v"(y6), "v"(y7));
        }
        // 8 fma_f64 + 4 ds_write_b64
        if constexpr (OP == 16) {
            asm volatile("ds_write_b64 %0, %1\nds_write_b64 %0, %2 offset:2048\nds_write_b64 %0, %3 offset:4096\nds_write_b64 %0, %4 offset:6144\n"
                         :: "v"(la), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory");
            asm volatile(R8(S0) OPS_D);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // mixed 4 fma_f64 + 4 v_mov_b32 (does a 32-bit op cost half an fp64 slot?)
        if constexpr (OP == 17)
            asm volatile("v_fma_f64 %0, %0, %8, %9\nv_mov_b32 %4, %10\nv_fma_f64 %1, %1, %8, %9\nv_mov_b32 %5, %10\n"
                         "v_fma_f64 %2, %2, %8, %9\nv_mov_b32 %6, %10\nv_fma_f64 %3, %3, %8, %9\nv_mov_b32 %7, %10\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib));
        // mixed 4 fma_f64 + 4 dpp moves
        if constexpr (OP == 18)
            asm volatile("v_fma_f64 %0, %0, %8, %9\nv_mov_b32_dpp %4, %10 row_shr:1 row_mask:0xf bank_mask:0xf\nv_fma_f64 %1, %1, %8, %9\nv_mov_b32_dpp %5, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fma_f64 %2, %2, %8, %9\nv_mov_b32_dpp %6, %10 row_shr:1 row_mask:0xf bank_mask:0xf\nv_fma_f64 %3, %3, %8, %9\nv_mov_b32_dpp %7, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib));
        // fma with two SGPR-free constant operands vs three VGPR operands: (register-read energy)
#define S19(n) "v_fma_f64 %" #n ", %" #n ", 2.0, 1.0\n"
        if constexpr (OP == 19) asm volatile(R8(S19) OPS_D);
#define S20(n) "v_mul_f64 %" #n ", %" #n ", %" #n "\n"
        if constexpr (OP == 20) asm volatile(R8(S20) OPS_D);
        // widening a float (the Float32 storage format): is v_cvt_f64_f32 a full-rate instruction?
#define S21(n) "v_cvt_f64_f32 %" #n ", %8\n"
        if constexpr (OP == 21)
            asm volatile(R8(S21) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(ib));
#define S22(n) "v_cvt_f32_f64 %" #n ", %8\n"
        if constexpr (OP == 22)
            asm volatile(R8(S22) : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7);
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
        rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r1 - r0;
    }
}

template <int OP>
void run(const char* name, int W, double seconds) {
    const int blocks = 256 * W;   // 256-thread blocks: 1 wave per SIMD per block
    double* out;
    unsigned long long *cyc, *rt;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    hipMalloc(&rt, sizeof(unsigned long long) * blocks * 4);
    auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    do {
        for (int j = 0; j < 20; ++j) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, rt, 1.0000001, 1e-9);
        hipDeviceSynchronize();
        launches += 20;
    } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
    std::vector<unsigned long long> h(blocks * 4), r(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), rt, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::vector<double> clk(blocks * 4);
    for (size_t i = 0; i < h.size(); ++i) clk[i] = (double)h[i] / (double)r[i] * 0.1;   // GHz
    std::sort(h.begin(), h.end());
    std::sort(clk.begin(), clk.end());
    const double med = (double)h[h.size() / 2];
    printf("%-22s W=%d  cycles/instr/SIMD=%6.3f  clock=%.3f GHz  -> %.3f ns per wave-instr  (%d launches)\n", name, W,
           med / (W * (double)ITER * 8), clk[clk.size() / 2], med / (W * (double)ITER * 8) / clk[clk.size() / 2], launches);
    fflush(stdout);
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 0.4;
    for (int w : {4}) {
        run<0>("v_fma_f64", w, secs); run<1>("v_mul_f64", w, secs); run<2>("v_add_f64", w, secs); run<3>("v_max_f64", w, secs);
        run<19>("v_fma_f64 consts", w, secs); run<20>("v_mul_f64 x,x,x", w, secs);
        run<4>("v_mov_b64", w, secs); run<5>("v_mov_b32", w, secs); run<6>("v_mov_b32 dpp row_shr", w, secs);
        run<7>("v_mov_b32 dpp wave_shr", w, secs); run<8>("v_rcp_f64", w, secs); run<9>("v_rsq_f64", w, secs);
        run<10>("v_pk_fma_f32", w, secs); run<11>("v_fma_f32", w, secs); run<12>("v_add_u32", w, secs);
        run<13>("ds_read_b64 x8", w, secs); run<14>("8 fma_f64 + 4 ds_read", w, secs); run<15>("8 fma_f64 + 8 ds_read", w, secs);
        run<21>("v_cvt_f64_f32", w, secs); run<22>("v_cvt_f32_f64", w, secs);
        run<16>("8 fma_f64 + 4 ds_write", w, secs); run<17>("4 fma_f64 + 4 mov_b32", w, secs); run<18>("4 fma_f64 + 4 dpp", w, secs);
    }
    for (int w : {1, 2, 6}) { run<0>("v_fma_f64", w, secs); run<5>("v_mov_b32", w, secs); run<14>("8 fma_f64 + 4 ds_read", w, secs); }
    return 0;
}
