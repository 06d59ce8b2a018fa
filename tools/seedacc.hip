// seedacc.hip — max relative error of the v_rcp_f64 / v_rsq_f64 seeds and of the Newton-refined
// values used by stage_math.h (FAST mode), against correctly rounded host results.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    o[i] = r;
    double e = __builtin_fma(-v, r, 1.0); double r1 = __builtin_fma(r, e, r);
    o[n + i] = r1;
    e = __builtin_fma(-v, r1, 1.0); o[2 * n + i] = __builtin_fma(r1, e, r1);
    double s = __builtin_amdgcn_rsq(v);
    o[3 * n + i] = s;
    double hx = 0.5 * v; e = __builtin_fma(-hx * s, s, 0.5); double s1 = __builtin_fma(s, e, s);
    o[4 * n + i] = s1;
    e = __builtin_fma(-hx * s1, s1, 0.5); o[5 * n + i] = __builtin_fma(s1, e, s1);
}
int main() {
    const int n = 1 << 20;
    std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-30, 30), m(1, 2);
    std::vector<double> x(n), o(6 * n);
    for (auto& v : x) v = m(g) * std::pow(10.0, u(g));
    double *dx, *dout; (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dout, 6 * n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    (void)hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    const char* names[6] = {"rcp seed", "rcp 1 NR", "rcp 2 NR", "rsq seed", "rsq 1 NR", "rsq 2 NR"};
    for (int j = 0; j < 6; ++j) {
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            long double ref = j < 3 ? 1.0L / x[i] : 1.0L / sqrtl((long double)x[i]);
            double err = (double)fabsl(((long double)o[j * n + i] - ref) / ref);
            if (err > worst) worst = err;
        }
        printf("%-10s max rel err = %.3e  (2^%.1f)\n", names[j], worst, std::log2(worst));
    }
    return 0;
}
