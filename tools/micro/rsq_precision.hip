// How accurate is v_rsq_f64, and after one / two Newton steps?  (decides how many steps rsqrt_pivot needs)
// hipcc --offload-arch=gfx950 -O3 tools/micro/rsq_precision.hip -o /tmp/rsqp && /tmp/rsqp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *d, double *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = d[i];
    double y0 = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    double y1 = y0 * fma(-h * y0, y0, 1.5);
    double y2 = y1 * fma(-h * y1, y1, 1.5);
    // one step, residual form: e = 1 - x y0^2 (fma), y = y0 + y0 * e / 2
    double e = fma(-x * y0, y0, 1.0);
    double y1r = fma(0.5 * y0, e, y0);
    o[4 * i] = y0; o[4 * i + 1] = y1; o[4 * i + 2] = y2; o[4 * i + 3] = y1r;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), o(4 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); h[i] = std::exp((u - 0.5) * 40.0); }
    double *dd, *dout;
    hipMalloc(&dd, n * 8); hipMalloc(&dout, 4 * n * 8);
    hipMemcpy(dd, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dd, dout, n);
    hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
    double m[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        long double ref = 1.0L / sqrtl((long double)h[i]);
        for (int q = 0; q < 4; ++q) { double e = (double)fabsl(((long double)o[4 * i + q] - ref) / ref); if (e > m[q]) m[q] = e; }
    }
    printf("max relative error: v_rsq_f64 %.3e, one Newton step %.3e, two steps %.3e, one step in residual form %.3e (ulp = 1.1e-16)\n", m[0], m[1], m[2], m[3]);
    return 0;
}
