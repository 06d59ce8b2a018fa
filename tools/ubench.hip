// ubench.hip — issue-cost microbenchmark of the VALU instructions the stage kernel is made of
// (gfx950).  Every wave runs ITER × 8 independent copies of ONE instruction between two s_memtime
// stamps; with W waves per SIMD resident the SIMD's sustained cycles/instruction is
// elapsed_cycles / (W·ITER·8).  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2048

#define BODY8(ASM)                                                                                  \
    asm volatile(ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM                  \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)   \
                 : "v"(b), "v"(c)                                                                   \
                 : "vcc");

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* cyc, double b, double c) {
    double x0 = threadIdx.x * 1e-3 + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    const int ib = (int)b, ib2 = (int)(c * 1e9);
    const unsigned long long smask = 0x5555555555555555ull + (unsigned long long)b;
    float f0 = i0, f1 = i1, f2 = i2, f3 = i3, f4 = i4, f5 = i5, f6 = i6, f7 = i7;
    const float fb = (float)b;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
        if constexpr (OP == 0)
            asm volatile("v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %1, %1, %8, %9\nv_fma_f64 %2, %2, %8, %9\nv_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\nv_fma_f64 %5, %5, %8, %9\nv_fma_f64 %6, %6, %8, %9\nv_fma_f64 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 1)
            asm volatile("v_mul_f64 %0, %0, %8\nv_mul_f64 %1, %1, %8\nv_mul_f64 %2, %2, %8\nv_mul_f64 %3, %3, %8\n"
                         "v_mul_f64 %4, %4, %8\nv_mul_f64 %5, %5, %8\nv_mul_f64 %6, %6, %8\nv_mul_f64 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 2)
            asm volatile("v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\n"
                         "v_add_f64 %4, %4, %8\nv_add_f64 %5, %5, %8\nv_add_f64 %6, %6, %8\nv_add_f64 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 3)
            asm volatile("v_max_f64 %0, %0, %8\nv_max_f64 %1, %1, %8\nv_max_f64 %2, %2, %8\nv_max_f64 %3, %3, %8\n"
                         "v_max_f64 %4, %4, %8\nv_max_f64 %5, %5, %8\nv_max_f64 %6, %6, %8\nv_max_f64 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 4)   // 32-bit select (low dword of each accumulator)
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib), "v"(c) : "vcc");
        if constexpr (OP == 5)
            asm volatile("v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\n"
                         "v_xor_b32 %4, %4, %8\nv_xor_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_xor_b32 %7, %7, %8"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib), "v"(c));
        if constexpr (OP == 6)
            asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3\n"
                         "v_rcp_f64 %4, %4\nv_rcp_f64 %5, %5\nv_rcp_f64 %6, %6\nv_rcp_f64 %7, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 7)
            asm volatile("v_rsq_f64 %0, %0\nv_rsq_f64 %1, %1\nv_rsq_f64 %2, %2\nv_rsq_f64 %3, %3\n"
                         "v_rsq_f64 %4, %4\nv_rsq_f64 %5, %5\nv_rsq_f64 %6, %6\nv_rsq_f64 %7, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 8)
            asm volatile("v_cmp_gt_f64 vcc, %0, %8\nv_cmp_gt_f64 vcc, %1, %8\nv_cmp_gt_f64 vcc, %2, %8\nv_cmp_gt_f64 vcc, %3, %8\n"
                         "v_cmp_gt_f64 vcc, %4, %8\nv_cmp_gt_f64 vcc, %5, %8\nv_cmp_gt_f64 vcc, %6, %8\nv_cmp_gt_f64 vcc, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c) : "vcc");
        if constexpr (OP == 9)   // fp32 reference
            asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fb), "v"(c));
        if constexpr (OP == 10)
            asm volatile("v_ldexp_f64 %0, %0, -2\nv_ldexp_f64 %1, %1, -2\nv_ldexp_f64 %2, %2, -2\nv_ldexp_f64 %3, %3, -2\n"
                         "v_ldexp_f64 %4, %4, -2\nv_ldexp_f64 %5, %5, -2\nv_ldexp_f64 %6, %6, -2\nv_ldexp_f64 %7, %7, -2"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 11)   // 64-bit move (register copy)
            asm volatile("v_mov_b64 %0, %8\nv_mov_b64 %1, %8\nv_mov_b64 %2, %8\nv_mov_b64 %3, %8\n"
                         "v_mov_b64 %4, %8\nv_mov_b64 %5, %8\nv_mov_b64 %6, %8\nv_mov_b64 %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 12)   // dependent fp64 chain (latency)
            asm volatile("v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %0, %0, %8, %9\nv_fma_f64 %0, %0, %8, %9\nv_fma_f64 %0, %0, %8, %9\n"
                         "v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %0, %0, %8, %9\nv_fma_f64 %0, %0, %8, %9\nv_fma_f64 %0, %0, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 13)   // fp64 fma with an SGPR/literal-free neg modifier + abs (modifiers are free?)
            asm volatile("v_fma_f64 %0, -%0, |%8|, %9\nv_fma_f64 %1, -%1, |%8|, %9\nv_fma_f64 %2, -%2, |%8|, %9\nv_fma_f64 %3, -%3, |%8|, %9\n"
                         "v_fma_f64 %4, -%4, |%8|, %9\nv_fma_f64 %5, -%5, |%8|, %9\nv_fma_f64 %6, -%6, |%8|, %9\nv_fma_f64 %7, -%7, |%8|, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 15)
            asm volatile("v_bfi_b32 %0, %8, %0, %9\nv_bfi_b32 %1, %8, %1, %9\nv_bfi_b32 %2, %8, %2, %9\nv_bfi_b32 %3, %8, %3, %9\n"
                         "v_bfi_b32 %4, %8, %4, %9\nv_bfi_b32 %5, %8, %5, %9\nv_bfi_b32 %6, %8, %6, %9\nv_bfi_b32 %7, %8, %7, %9"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib), "v"(ib2));
        if constexpr (OP == 16)
            asm volatile("v_ashrrev_i32 %0, 31, %0\nv_ashrrev_i32 %1, 31, %1\nv_ashrrev_i32 %2, 31, %2\nv_ashrrev_i32 %3, 31, %3\n"
                         "v_ashrrev_i32 %4, 31, %4\nv_ashrrev_i32 %5, 31, %5\nv_ashrrev_i32 %6, 31, %6\nv_ashrrev_i32 %7, 31, %7"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib), "v"(ib2));
        if constexpr (OP == 17)   // realistic 64-bit select: 4 x (v_cmp_gt_f64 vcc ; 2 x v_cndmask)
            asm volatile("v_cmp_gt_f64 vcc, %0, %8\nv_cndmask_b32 %4, %4, %9, vcc\nv_cndmask_b32 %5, %5, %9, vcc\n"
                         "v_cmp_gt_f64 vcc, %1, %8\nv_cndmask_b32 %6, %6, %9, vcc\nv_cndmask_b32 %7, %7, %9, vcc\n"
                         "v_cmp_gt_f64 vcc, %2, %8\nv_cndmask_b32 %4, %4, %9, vcc\nv_cndmask_b32 %5, %5, %9, vcc\n"
                         "v_cmp_gt_f64 vcc, %3, %8\nv_cndmask_b32 %6, %6, %9, vcc\nv_cndmask_b32 %7, %7, %9, vcc"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(ib) : "vcc");
        if constexpr (OP == 18)
            asm volatile("v_lshl_add_u64 %0, %0, 3, %8\nv_lshl_add_u64 %1, %1, 3, %8\nv_lshl_add_u64 %2, %2, 3, %8\nv_lshl_add_u64 %3, %3, 3, %8\n"
                         "v_lshl_add_u64 %4, %4, 3, %8\nv_lshl_add_u64 %5, %5, 3, %8\nv_lshl_add_u64 %6, %6, 3, %8\nv_lshl_add_u64 %7, %7, 3, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 19)
            asm volatile("v_min_f64 %0, |%0|, |%8|\nv_min_f64 %1, |%1|, |%8|\nv_min_f64 %2, |%2|, |%8|\nv_min_f64 %3, |%3|, |%8|\n"
                         "v_min_f64 %4, |%4|, |%8|\nv_min_f64 %5, |%5|, |%8|\nv_min_f64 %6, |%6|, |%8|\nv_min_f64 %7, |%7|, |%8|"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c));
        if constexpr (OP == 20)   // mixed: fma_f64 interleaved with cndmask (vcc) 1:1
            asm volatile("v_fma_f64 %0, %0, %8, %8\nv_cndmask_b32 %4, %4, %9, vcc\nv_fma_f64 %1, %1, %8, %8\nv_cndmask_b32 %5, %5, %9, vcc\n"
                         "v_fma_f64 %2, %2, %8, %8\nv_cndmask_b32 %6, %6, %9, vcc\nv_fma_f64 %3, %3, %8, %8\nv_cndmask_b32 %7, %7, %9, vcc"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(ib) : "vcc");
        if constexpr (OP == 21)
            asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib), "v"(ib2));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7) + (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 1 wave per SIMD per block
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001, 1e-9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    // per SIMD: waves_per_simd waves × ITER × 8 instructions in `med` counter ticks (per-wave elapsed)
    printf("%-14s W=%d  ticks/wave=%.0f  ticks per instr per SIMD=%.3f  wall=%.3f ms  (instr/s/SIMD -> %.3f ns each)\n", name,
           waves_per_simd, med, med / (waves_per_simd * (double)ITER * 8), ms,
           ms * 1e6 / (waves_per_simd * (double)ITER * 8));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {2, 4}) {
        run<0>("v_fma_f64", w); run<1>("v_mul_f64", w); run<2>("v_add_f64", w); run<3>("v_max_f64", w);
        run<4>("v_cndmask_b32", w); run<5>("v_xor_b32", w); run<6>("v_rcp_f64", w); run<7>("v_rsq_f64", w);
        run<8>("v_cmp_gt_f64", w); run<9>("v_fma_f32", w); run<10>("v_ldexp_f64", w); run<11>("v_mov_b64", w);
        run<12>("fma_f64 chain", w); run<13>("fma_f64 +mods", w);
        run<15>("v_bfi_b32", w); run<16>("v_ashrrev_i32", w); run<17>("cmp+2cndmask x4", w);
        run<18>("v_lshl_add_u64", w); run<19>("v_min_f64 |abs|", w); run<20>("fma+cndmask 1:1", w); run<21>("v_add_u32", w);
    }
    return 0;
}
