// Microbenchmark: v_mfma_f64_16x16x4_f64 issue rate on gfx950 (confirms the fp64 matrix peak used
// by bench.py's roofline).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double *out, int iters, long long *cyc)
{
    d4 acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int blocks, int threads, int iters)
{
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double nm = (double)blocks * (threads / 64) * iters * NACC;
    double waves_per_simd = (double)blocks * (threads / 64) / 1024.0;
    printf("NACC=%d blocks=%d threads=%d (%.0f waves/SIMD): %.3f ms, %.2f TFLOP/s fp64, %.1f ns per MFMA per wave, s_memtime %.1f ticks per MFMA (block 0)\n",
           NACC, blocks, threads, waves_per_simd, ms, nm * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * NACC), (double)c / (iters * NACC));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<1>(256, 256, 20000);       // one dependent chain per wave, ONE wave per SIMD: exposes the dependent-issue latency
    run<2>(256, 256, 20000);
    run<3>(256, 256, 20000);
    run<1>(256 * 4, 256, 20000);   // one dependent chain per wave, 4 waves per SIMD
    run<4>(256, 256, 20000);       // 4 independent accumulators, one wave per SIMD
    run<4>(256 * 2, 256, 20000);   // two waves per SIMD
    run<8>(256 * 2, 256, 20000);
    return 0;
}
