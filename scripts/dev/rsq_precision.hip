// Probe: relative error of v_rsq_f64 / v_rcp_f64 raw, after one and after two Newton steps (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rsq_probe scripts/dev/rsq_precision.hip && /tmp/rsq_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void probe(const double *x, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r0 = __builtin_amdgcn_rsq(d);
    double hd = 0.5 * d;
    double e = fma(-hd * r0, r0, 0.5);
    double r1 = fma(r0, e, r0);
    e = fma(-hd * r1, r1, 0.5);
    double r2 = fma(r1, e, r1);
    double c0 = __builtin_amdgcn_rcp(d);
    double c1 = fma(c0, fma(-d, c0, 1.0), c0);
    double c2 = fma(c1, fma(-d, c1, 1.0), c1);
    // the single third-order corrections nnls_wave.hpp uses
    double e3 = fma(-hd * r0, r0, 0.5);
    double r3 = fma(r0, e3 * fma(1.5, e3, 1.0), r0);
    double f3 = fma(-d, c0, 1.0);
    double c3 = fma(c0, fma(f3, f3, f3), c0);
    out[8 * i + 0] = r0; out[8 * i + 1] = r1; out[8 * i + 2] = r2; out[8 * i + 6] = r3;
    out[8 * i + 3] = c0; out[8 * i + 4] = c1; out[8 * i + 5] = c2; out[8 * i + 7] = c3;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(8 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 8 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 8 * n * 8, hipMemcpyDeviceToHost);
    double mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / sqrtl((long double)x[i]), c = 1.0L / (long double)x[i];
        const int isr[8] = {1, 1, 1, 0, 0, 0, 1, 0};
        for (int k = 0; k < 8; ++k) { long double ref = isr[k] ? t : c; double er = fabs((double)(((long double)o[8 * i + k] - ref) / ref)); if (er > mx[k]) mx[k] = er; }
    }
    printf("max relative error over %d values (eps = 2.2e-16):\n rsq raw %.3e  +1 Newton %.3e  +2 Newton %.3e  one 3rd-order step %.3e\n rcp raw %.3e  +1 Newton %.3e  +2 Newton %.3e  one 3rd-order step %.3e\n", n, mx[0], mx[1], mx[2], mx[6], mx[3], mx[4], mx[5], mx[7]);
    return 0;
}
