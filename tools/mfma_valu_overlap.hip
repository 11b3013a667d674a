// Microbenchmark: do fp64 MFMA and fp64 / fp32 VALU work from DIFFERENT waves of one SIMD overlap on gfx950?
// One workgroup of 8 waves per CU (two waves per SIMD).  Waves 0-3 run role A, waves 4-7 role B; HW_ID is
// recorded so the pairing per SIMD can be checked.  Modes: A alone, B alone, both.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// role 0 idle, 1 = f64 MFMA (4 independent accumulators), 2 = f64 FMA (8 independent chains),
// 3 = f32 FMA (8 chains), 4 = f64 MFMA single dependent chain, 5 = f64 FMA single dependent chain
__device__ __forceinline__ double work(int role, int iters)
{
    double s = 0;
    if (role == 1) {
        d4 acc[4] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    } else if (role == 4) {
        d4 acc = d4{0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int i = 0; i < iters * 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        s = acc[0] + acc[1] + acc[2] + acc[3];
    } else if (role == 2) {
        double x[8];
        for (int q = 0; q < 8; ++q) x[q] = threadIdx.x * 1e-3 + q;
        const double m = 0.999999, c = 1e-7;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r)           // 64 FMA instructions per iteration
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fma(x[q], m, c);
        for (int q = 0; q < 8; ++q) s += x[q];
    } else if (role == 5) {
        double x = threadIdx.x * 1e-3;
        const double m = 0.999999, c = 1e-7;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 64; ++r) x = fma(x, m, c);
        s = x;
    } else if (role == 3) {
        float x[8];
        for (int q = 0; q < 8; ++q) x[q] = threadIdx.x * 1e-3f + q;
        const float m = 0.99999f, c = 1e-6f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fmaf(x[q], m, c);
        for (int q = 0; q < 8; ++q) s += x[q];
    }
    return s;
}

__global__ __launch_bounds__(512) void k(double *out, int roleA, int roleB, int iters, long long *cyc, int *hw)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    __syncthreads();
    long long t0 = clock64();
    double s = work(role, iters);
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * 8 + wave] = t1 - t0;
        hw[blockIdx.x * 8 + wave] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

static const char *names[] = {"idle", "f64 MFMA x4acc", "f64 FMA x8", "f32 FMA x8", "f64 MFMA chain", "f64 FMA chain"};

void run(int roleA, int roleB, int iters)
{
    const int blocks = 256;
    double *out; long long *cyc; int *hw;
    hipMalloc(&out, sizeof(double) * blocks * 512);
    hipMalloc(&cyc, sizeof(long long) * blocks * 8);
    hipMalloc(&hw, sizeof(int) * blocks * 8);
    k<<<blocks, 512>>>(out, roleA, roleB, 10, cyc, hw);
    hipDeviceSynchronize();
    k<<<blocks, 512>>>(out, roleA, roleB, iters, cyc, hw);
    hipDeviceSynchronize();
    long long c[8]; int h[8];
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    hipMemcpy(h, hw, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on this part: report ticks per "unit" (one MFMA, or 16 FMA instructions = 64 issue cycles)
    auto per = [&](int role, long long t) { return role ? (double)t / (iters * 4.0) : 0.0; };
    printf("A=%-15s B=%-15s | A: %.3f ticks/unit  B: %.3f ticks/unit | SIMD of waves:", names[roleA], names[roleB],
           per(roleA, c[0]), per(roleB, c[4]));
    for (int w = 0; w < 8; ++w) printf(" %d", (h[w] >> 4) & 3);
    printf("\n");
    hipFree(out); hipFree(cyc); hipFree(hw);
}

int main()
{
    // unit = one MFMA (64 cycles of the matrix pipe) or 16 VALU FMA wave-instructions (64 issue cycles)
    const int it = 20000;
    run(1, 0, it); run(2, 0, it); run(3, 0, it); run(4, 0, it); run(5, 0, it);
    run(1, 2, it);     // f64 MFMA vs f64 VALU on the same SIMD
    run(1, 3, it);     // f64 MFMA vs f32 VALU
    run(1, 1, it);     // two MFMA waves
    run(2, 2, it);     // two f64 VALU waves
    run(4, 2, it);     // dependent MFMA chain vs f64 VALU throughput wave
    run(4, 5, it);     // dependent MFMA chain vs dependent FMA chain
    run(5, 5, it);
    return 0;
}
