// DIAGNOSTIC micro-benchmark: how long does the GPU take to dispatch G workgroups of T threads that do (almost) nothing?
// hipcc --offload-arch=gfx950 -O3 -o dispatch_rate dispatch_rate.hip && ./dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS, int SCR>
__global__ void k_empty(double *out, int spin)
{
    __shared__ double s[LDS > 0 ? LDS : 1];
    double loc[SCR > 0 ? SCR : 1];
    if (LDS > 0) s[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < (SCR > 0 ? SCR : 1); ++i) loc[i] = i + spin;
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    double acc = 0.0;
    for (int i = 0; i < (SCR > 0 ? SCR : 1); ++i) acc += loc[(i * 7 + spin) % (SCR > 0 ? SCR : 1)];
    if (out && spin < 0) out[blockIdx.x] = acc + (LDS > 0 ? s[(threadIdx.x + 1) % LDS] : 0.0);
}
template <class K>
static void run(const char *name, K kern, int G, int T, int spin)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(T), 0, 0, nullptr, spin);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int R = 50;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(T), 0, 0, nullptr, spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s G=%5d T=%4d spin=%6d cycles: %.2f us per launch\n", name, G, T, spin, 1000.0 * ms / R);
}
int main()
{
    for (int spin : {0, 10000, 20000}) {
        run("no LDS, no scratch", k_empty<0, 0>, 4096, 64, spin);
        run("8 KB LDS", k_empty<1024, 0>, 4096, 64, spin);
        run("scratch (32 doubles, dynamic index)", k_empty<0, 32>, 4096, 64, spin);
        run("no LDS, no scratch", k_empty<0, 0>, 1024, 256, spin);
        run("8 KB LDS", k_empty<1024, 0>, 1024, 256, spin);
        run("scratch", k_empty<0, 32>, 1024, 256, spin);
        run("no LDS, no scratch", k_empty<0, 0>, 16384, 64, spin);
    }
    return 0;
}
