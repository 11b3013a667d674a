// slk_api.hip -- host side of the C ABI declared in include/slk.h.
// Owns the batch's device buffers, stages host arguments, picks the kernel instantiation and
// launches on the handle's stream.  There is NO CPU fallback: without a HIP device every
// entry point fails with SLK_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/slk.h"
#include "slk_kernels.hpp"
#include "slk_usckf.hpp"
#include "slk_ekf.hpp"
#include "slk_ekf_tiles.hpp"
#include "slk_pose.hpp"

// The largest step-kernel instantiations are compiled in translation units of their own (slk_inst_big.hip,
// slk_inst_mid.hip) so that the library's build runs them in parallel; development builds (one file) keep none of them.
#if !defined(SLK_DEV_N60) && !defined(SLK_ONE_TU)
namespace slk {
extern template __global__ void msckf_step_kernel<13, 512, -1, 0>(KArgs);
extern template __global__ void msckf_step_kernel<13, 512, 31, 8>(KArgs);
extern template __global__ void msckf_step_kernel<10, 512, -1, 0>(KArgs);
extern template __global__ void msckf_step_kernel<8, 256, -1, 0>(KArgs);
extern template __global__ void msckf_step_kernel<6, 256, -1, 0>(KArgs);
extern template __global__ void msckf_step_kernel<5, 256, -1, 0>(KArgs);
} // namespace slk
#endif

using namespace slk;

static thread_local std::string g_err;

#define HIPCHECK(expr)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return SLK_E_HIP;                                                            \
        }                                                                                \
    } while (0)

struct Stage {
    double *p = nullptr;
    size_t cap = 0;
};

struct slk_filter {
    slk_config cfg;
    Lay lay;
    int B;
    hipStream_t stream;
    bool own_stream;
    double *d_mean, *d_P;
    size_t cap_mean, cap_P;       // capacities of d_mean / d_P in doubles (whole batch)
    double *d_mean_alt = nullptr, *d_P_alt = nullptr;   // second pair of state buffers: layout changes (window push / pop,
    size_t cap_mean_alt = 0, cap_P_alt = 0;             // setMeasurement) are built into it on the stream and swapped in
    int *d_status;
    unsigned *d_outliers;
    Stage st_u, st_Q, st_mp, st_z, st_R, st_X, st_Z, st_tmpP, st_tmpM;
    Stage ws_L, ws_DR;            // large-state workspaces (N > 80), allocated on first use
    Stage ws_ekf;                 // EKF update workspace, allocated on first use
    // Msckf rotation-item descriptors, one table per window length k the handle has run (a sliding window alternates
    // between k and k + 1: the tables stay, so the steady state allocates and synchronises nothing)
    struct Rtab { unsigned long long *dev = nullptr; std::vector<unsigned long long> host; };
    std::map<int, Rtab> rtabs;
    hipEvent_t ev0, ev1;
    int rebuild_prec = 0;
    // The exact-shape Msckf update kernels store P+ as lower triangle + diagonal tiles (KArgs::lower_only); the strict upper
    // triangle is brought up to date (mirror_upper) before anything but those kernels, predict and the factor kernels --
    // which read the lower triangle only -- gets to see the matrix.
    bool upper_stale = false;
};

static int mirror_upper(slk_filter *f);

static Lay make_lay(int kind, int k, int nfk, int nfkl)
{
    Lay L;
    L.kind = kind; L.k = k; L.nfk = nfk; L.nfkl = nfkl;
    if (kind == SLK_MSCKF) { L.N = 12 + 6 * k; L.Nq = 13 + 7 * k; L.nso3 = 1 + k; }
    else { L.N = 36 + nfk + nfkl; L.Nq = 39 + nfk + nfkl; L.nso3 = 3; }
    return L;
}

static int stage_reserve(slk_filter *f, Stage &s, size_t n)
{
    if (s.cap >= n) return SLK_OK;
    if (s.p) HIPCHECK(hipFree(s.p));
    s.p = nullptr; s.cap = 0;
    HIPCHECK(hipMalloc(&s.p, n * sizeof(double)));
    s.cap = n;
    (void)f;
    return SLK_OK;
}

// returns a device pointer for `src` (n doubles): in place for SLK_DEVICE, staged copy for SLK_HOST
static int stage_in(slk_filter *f, Stage &s, const double *src, size_t n, int where, const double **out)
{
    if (!src || n == 0) { *out = nullptr; return SLK_OK; }
    if (where == SLK_DEVICE) { *out = src; return SLK_OK; }
    int rc = stage_reserve(f, s, n);
    if (rc) return rc;
    HIPCHECK(hipMemcpyAsync(s.p, src, n * sizeof(double), hipMemcpyHostToDevice, f->stream));
    *out = s.p;
    return SLK_OK;
}

extern "C" {

const char *slk_last_error(void) { return g_err.c_str(); }

int slk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int slk_create(const slk_config *cfg, slk_filter **out)
{
    if (!cfg || !out) return SLK_E_INVALID;
    if (cfg->kind != SLK_MSCKF && cfg->kind != SLK_USCKF) return SLK_E_INVALID;
    if (cfg->batch < 1 || cfg->n_clones < 0 || cfg->n_featuresk < 0 || cfg->n_featuresk_l < 0) return SLK_E_INVALID;
    int ndev = slk_device_count();
    if (ndev <= 0) { g_err = "no HIP device: the slk library has no CPU fallback"; return SLK_E_NO_DEVICE; }
    if (cfg->device < 0 || cfg->device >= ndev) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(cfg->device));
    slk_filter *f = new slk_filter();
    f->cfg = *cfg;
    f->lay = make_lay(cfg->kind, cfg->n_clones, cfg->n_featuresk, cfg->n_featuresk_l);
    f->B = cfg->batch;
    f->own_stream = (cfg->stream == nullptr);
    if (f->own_stream) {
        hipError_t e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { g_err = hipGetErrorString(e); delete f; return SLK_E_HIP; }
    } else {
        f->stream = (hipStream_t)cfg->stream;
    }
    f->d_mean = f->d_P = nullptr;
    f->d_status = nullptr; f->d_outliers = nullptr;
    size_t B = (size_t)f->B;
    f->cap_mean = B * (size_t)f->lay.Nq; f->cap_P = B * (size_t)f->lay.N * f->lay.N;
    bool ok = hipMalloc(&f->d_mean, f->cap_mean * sizeof(double)) == hipSuccess
           && hipMalloc(&f->d_P, f->cap_P * sizeof(double)) == hipSuccess
           && hipMalloc(&f->d_status, B * sizeof(int)) == hipSuccess
           && hipMalloc(&f->d_outliers, B * sizeof(unsigned)) == hipSuccess
           && hipEventCreate(&f->ev0) == hipSuccess && hipEventCreate(&f->ev1) == hipSuccess;
    if (!ok) { g_err = "device allocation failed"; slk_destroy(f); return SLK_E_NOMEM; }
    (void)hipMemsetAsync(f->d_status, 0, B * sizeof(int), f->stream);
    (void)hipMemsetAsync(f->d_outliers, 0, B * sizeof(unsigned), f->stream);
    (void)hipMemsetAsync(f->d_mean, 0, f->cap_mean * sizeof(double), f->stream);
    (void)hipMemsetAsync(f->d_P, 0, f->cap_P * sizeof(double), f->stream);
    *out = f;
    return SLK_OK;
}

void slk_destroy(slk_filter *f)
{
    if (!f) return;
    (void)hipSetDevice(f->cfg.device);
    (void)hipStreamSynchronize(f->stream);
    Stage *st[] = {&f->st_u, &f->st_Q, &f->st_mp, &f->st_z, &f->st_R, &f->st_X, &f->st_Z, &f->st_tmpP, &f->st_tmpM,
                   &f->ws_L, &f->ws_DR, &f->ws_ekf};
    for (Stage *s : st) if (s->p) (void)hipFree(s->p);
    for (auto &kv : f->rtabs) if (kv.second.dev) (void)hipFree(kv.second.dev);
    if (f->d_mean) (void)hipFree(f->d_mean);
    if (f->d_P) (void)hipFree(f->d_P);
    if (f->d_mean_alt) (void)hipFree(f->d_mean_alt);
    if (f->d_P_alt) (void)hipFree(f->d_P_alt);
    if (f->d_status) (void)hipFree(f->d_status);
    if (f->d_outliers) (void)hipFree(f->d_outliers);
    (void)hipEventDestroy(f->ev0);
    (void)hipEventDestroy(f->ev1);
    if (f->own_stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

int slk_batch(const slk_filter *f) { return f ? f->B : SLK_E_INVALID; }
int slk_dof(const slk_filter *f) { return f ? f->lay.N : SLK_E_INVALID; }
int slk_storage(const slk_filter *f) { return f ? f->lay.Nq : SLK_E_INVALID; }
double *slk_mean_device_ptr(slk_filter *f) { return f ? f->d_mean : nullptr; }
double *slk_cov_device_ptr(slk_filter *f)
{
    if (!f || mirror_upper(f) != SLK_OK) return nullptr;      // (enqueued on the handle's stream, like every step)
    return f->d_P;
}

int slk_set_state(slk_filter *f, const double *mean, const double *P, int where)
{
    if (!f) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    hipMemcpyKind kind = where == SLK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    size_t B = (size_t)f->B;
    if (mean) HIPCHECK(hipMemcpyAsync(f->d_mean, mean, B * f->lay.Nq * sizeof(double), kind, f->stream));
    if (P) {
        HIPCHECK(hipMemcpyAsync(f->d_P, P, B * f->lay.N * f->lay.N * sizeof(double), kind, f->stream));
        f->upper_stale = false;
    }
    if (where == SLK_HOST) HIPCHECK(hipStreamSynchronize(f->stream));   // caller may reuse its buffers
    return SLK_OK;
}

int slk_get_state(slk_filter *f, double *mean, double *P, int where)
{
    if (!f) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    hipMemcpyKind kind = where == SLK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    size_t B = (size_t)f->B;
    if (mean) HIPCHECK(hipMemcpyAsync(mean, f->d_mean, B * f->lay.Nq * sizeof(double), kind, f->stream));
    if (P) { int rc = mirror_upper(f); if (rc) return rc; }
    if (P) HIPCHECK(hipMemcpyAsync(P, f->d_P, B * f->lay.N * f->lay.N * sizeof(double), kind, f->stream));
    if (where == SLK_HOST) HIPCHECK(hipStreamSynchronize(f->stream));
    return SLK_OK;
}

} // extern "C"

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function object of ONE device: remember what was
// configured per (kernel, device) -- a process may own handles on several GPUs (slk_config.device) and launch from
// several host threads.
static int ensure_dynamic_lds(const void *kern, int device, size_t lds)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> configured;
    std::lock_guard<std::mutex> guard(mu);
    size_t &have = configured[std::make_pair(kern, device)];
    if (lds > have) {
        HIPCHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        have = lds;
    }
    return SLK_OK;
}

// ---------------------------------------------------------------------------- launch helpers
template <int NT, int NTHREADS, int KST = -1, int MST = 0>
static int launch_msckf_inst(slk_filter *f, const KArgs &a0)
{
    KArgs a = a0;
    constexpr bool BIG = NT > 4;
    Carve cv = carve_step(a.lay, a.m, NT, BIG, a.rebuild_prec);
    {
        slk_filter::Rtab &rt = f->rtabs[a.lay.k];
        if (!rt.dev) {                             // first step at this window length: build the table, copy it on the stream
            rt.host.resize((size_t)cv.W);          // (the host copy lives as long as the handle: the copy needs no wait)
            for (int w = 0; w < cv.W; ++w) rt.host[w] = rot_item_descriptor(a.lay.N, w);
            HIPCHECK(hipMalloc(&rt.dev, rt.host.size() * sizeof(unsigned long long)));
            HIPCHECK(hipMemcpyAsync(rt.dev, rt.host.data(), rt.host.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, f->stream));
        }
        a.rtab = rt.dev;
    }
    if constexpr (NT >= 3 && NT <= 4) {
        // Three launches per step: predict (one wave per filter), the first factorisation (its own residency, the packed
        // factor handed over through a workspace), update + applyDelta.
        int rc = stage_reserve(f, f->ws_L, (size_t)a.B * pk_size(a.lay.N));
        if (rc) return rc;
        rc = stage_reserve(f, f->ws_DR, ((size_t)a.B * sizeof(int) + sizeof(double) - 1) / sizeof(double));
        if (rc) return rc;
        a.wsL = f->ws_L.p;
        a.wsfail = reinterpret_cast<int *>(f->ws_DR.p);
        size_t lds = (size_t)cv.total * sizeof(double);
        if (HasFastStep<NT, NTHREADS, KST, MST>::value) {       // the exact-shape fast path carves LDS its own way
            const size_t fl = (size_t)fast_step_lds_doubles(a.lay.k) * sizeof(double);
            if (fl > lds) lds = fl;
            if (a.do_update && a.emit == 0 && !a.P_out && a.P == f->d_P) {      // P+ as lower triangle + diagonal tiles (mirror_upper)
                a.lower_only = 1;
                f->upper_stale = true;
            }
#ifndef SLK_MSCKF_FACTOR_KERNEL      // the first factorisation inside the update kernel's fast path: two launches per step, no factor round trip
            if (a.do_update && a.emit == 0 && a.mm == SLK_MM_FEATURE_PROJ && a.m == 8 && a.gate != 2 && a.rebuild_prec == 0 && a.mp) a.wsfail = nullptr;
#endif
        }
        auto kern = msckf_step_kernel<NT, NTHREADS, KST, MST>;
        rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), f->cfg.device, lds);
        if (rc) return rc;
        auto run_part = [&](KArgs s, hipStream_t st) -> int {
            if (s.do_predict) {
                hipLaunchKernelGGL(msckf_predict_kernel, dim3(s.B), dim3(64), 0, st, s);
                HIPCHECK(hipGetLastError());
                s.do_predict = 0;                   // the step kernel takes the predicted state from memory
            }
            if (s.wsfail) {
                hipLaunchKernelGGL((msckf_chol_kernel<NT, KST>), dim3(s.B), dim3(64), 0, st, s);
                HIPCHECK(hipGetLastError());
            }
            hipLaunchKernelGGL(kern, dim3(s.B), dim3(NTHREADS), lds, st, s);
            HIPCHECK(hipGetLastError());
            return SLK_OK;
        };
        return run_part(a, f->stream);
    }
    if (BIG) {
        int rc = stage_reserve(f, f->ws_L, (size_t)a.B * pk_size(a.lay.N));
        if (rc) return rc;
        const size_t ndr = (size_t)a.B * 3 * cv.W;                   // rotation deviations, then one int per filter
        rc = stage_reserve(f, f->ws_DR, ndr + ((size_t)a.B * sizeof(int) + sizeof(double) - 1) / sizeof(double));
        if (rc) return rc;
        a.wsL = f->ws_L.p;
        a.wsDR = f->ws_DR.p;
        if (a.do_update || a.emit >= 2) {                            // the first factorisation in its own launch
            a.wsfail = reinterpret_cast<int *>(f->ws_DR.p + ndr);
            auto ck = msckf_chol_big_kernel<NTHREADS>;
            const size_t clds = chol_big_lds(a.lay.N);
            rc = ensure_dynamic_lds(reinterpret_cast<const void *>(ck), f->cfg.device, clds);
            if (rc) return rc;
            hipLaunchKernelGGL(ck, dim3(a.B), dim3(NTHREADS), clds, f->stream, a);
            HIPCHECK(hipGetLastError());
        }
    }
    size_t lds = (size_t)cv.total * sizeof(double);
    if (lds > 160 * 1024) { g_err = "state too large for the LDS-resident kernel"; return SLK_E_UNSUPPORTED; }
    auto kern = msckf_step_kernel<NT, NTHREADS, KST, MST>;
    int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), f->cfg.device, lds);
    if (rc_lds) return rc_lds;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(NTHREADS), lds, f->stream, a);
    HIPCHECK(hipGetLastError());
    return SLK_OK;
}

// Exact-shape instantiations of the headline workload (BASELINE.json configs[2] / [3]: k = 8 clones, N = 60; with
// m = 8 measurement rows every LDS offset is a compile-time constant too); every other shape runs the same kernel
// source with run-time sizes.
static int launch_msckf_n60(slk_filter *f, const KArgs &a)
{
    const bool m8 = a.m == 8 && a.do_update && a.emit == 0 && a.rebuild_prec == 0;
    if (a.lay.k == 8 && m8) return launch_msckf_inst<4, 256, 8, 8>(f, a);
    if (a.lay.k == 8) return launch_msckf_inst<4, 256, 8>(f, a);
#ifndef SLK_DEV_N60
    if (a.lay.k == 7 && m8) return launch_msckf_inst<4, 256, 7, 8>(f, a);       // the window on its way to eight clones
#endif
    return launch_msckf_inst<4, 256>(f, a);
}
#ifndef SLK_DEV_N60
static int launch_msckf_n48(slk_filter *f, const KArgs &a)                          // N = 36, 42, 48 (k = 4, 5, 6)
{
    const bool m8 = a.m == 8 && a.do_update && a.emit == 0 && a.rebuild_prec == 0;
    if (a.lay.k == 6 && m8) return launch_msckf_inst<3, 256, 6, 8>(f, a);
    if (a.lay.k == 5 && m8) return launch_msckf_inst<3, 256, 5, 8>(f, a);
    if (a.lay.k == 4 && m8) return launch_msckf_inst<3, 256, 4, 8>(f, a);
    return launch_msckf_inst<3, 256>(f, a);
}
#endif

// Windows beyond the LDS-resident kernels (N > 208; the reference's MultiState is unbounded, State.hpp:342, :373-376): the
// plain global-workspace kernel of slk_general.hpp -- slow, but every legal call works.
static int launch_msckf_general(slk_filter *f, const KArgs &a0)
{
    KArgs a = a0;
    if (a.do_update && a.m > MAXM) { g_err = "more than 32 measurement rows per update are not supported"; return SLK_E_UNSUPPORTED; }
    const GenWs w = general_ws(a.lay.N, a.lay.Nq, a.lay.nso3, a.m > 0 ? a.m : 1);
    int rc = stage_reserve(f, f->ws_L, (size_t)a.B * w.total);
    if (rc) return rc;
    a.wsL = f->ws_L.p;
    hipLaunchKernelGGL(msckf_update_general_kernel, dim3(a.B), dim3(256), 0, f->stream, a);
    HIPCHECK(hipGetLastError());
    return SLK_OK;
}

static int launch_msckf(slk_filter *f, const KArgs &a0)
{
    KArgs a = a0;
    int NT = (a.lay.N + 15) / 16;
    // predict / update / step read the lower triangle of P only (Msckf.hpp:412, :447); everything else gets the whole matrix
    if (f->upper_stale && (a.emit != 0 || a.P_out)) { int rc = mirror_upper(f); if (rc) return rc; }
    // fused step: NT 3 / 4 launch predict next to the factor kernel themselves, the one-wave kernels (N <= 32) run it inside
    // -- for small batches, where the step time is one filter's latency (B = 1024: 30.5 -> 28.0 us); large batches run the
    // predict chain in its own launch at its own residency (B = 16384: 79.5 against 70.2 M steps/s)
    const bool inside = (NT >= 3 ? NT <= 4 : a.B <= 4096) && a.do_predict && a.do_update && a.emit == 0;
    if ((a.do_predict || a.emit == 1) && !inside) {   // predict (or its Tier-B sigma-point emission): one wave per filter
        hipLaunchKernelGGL(msckf_predict_kernel, dim3(a.B), dim3(64), 0, f->stream, a);
        HIPCHECK(hipGetLastError());
        if (!a.do_update) return SLK_OK;
        a.do_predict = 0;                       // the step kernel takes the predicted state from memory
    }
#ifdef SLK_DEV_N60      // development builds (tools/ab.sh): only the headline instantiations, for quick A/B turnarounds
#ifdef SLK_DEV_SMALL    // (-DSLK_DEV_SMALL: only BASELINE config 2's exact shapes, N = 12 / m = 3 and N = 18 / m = 2)
    if (NT == 1 && a.lay.k == 0 && a.m == 3) return launch_msckf_inst<1, 64, 0, 3>(f, a);
    if (NT == 2 && a.lay.k == 1 && a.m == 2) return launch_msckf_inst<2, 64, 1, 2>(f, a);
#elif !defined(SLK_DEV_EKF)     // (-DSLK_DEV_EKF: only the EKF tile kernel)
    if (NT == 4) return launch_msckf_n60(f, a);
#endif
    g_err = "development build: N = 49..64 only"; return SLK_E_UNSUPPORTED;
#else
    switch (NT) {
    case 1: {                                   // N = 12 (BASELINE config 2): exact shape, with m = 3 rows exact offsets too
        const bool mx = a.do_update && a.emit == 0 && a.rebuild_prec == 0;
        if (a.lay.k == 0 && a.m == 3 && mx) return launch_msckf_inst<1, 64, 0, 3>(f, a);
        return (a.lay.k == 0) ? launch_msckf_inst<1, 64, 0>(f, a) : launch_msckf_inst<1, 64>(f, a);
    }
    case 2: {                                   // N = 18
        const bool mx = a.do_update && a.emit == 0 && a.rebuild_prec == 0;
        if (a.lay.k == 1 && a.m == 2 && mx) return launch_msckf_inst<2, 64, 1, 2>(f, a);
        return (a.lay.k == 1) ? launch_msckf_inst<2, 64, 1>(f, a) : launch_msckf_inst<2, 64>(f, a);
    }
    case 3: return launch_msckf_n48(f, a);
    case 4: return launch_msckf_n60(f, a);
    case 5: return launch_msckf_inst<5, 256>(f, a);
    case 6: return launch_msckf_inst<6, 256>(f, a);
    case 7: case 8: return launch_msckf_inst<8, 256>(f, a);
    case 9: case 10: return launch_msckf_inst<10, 512>(f, a);
    case 11: case 12: case 13:
        // BASELINE config 5 (N = 198): exact shape
        if (a.lay.k == 31 && a.m == 8 && a.do_update && a.emit == 0) return launch_msckf_inst<13, 512, 31, 8>(f, a);   // (any rebuild precision)
        return launch_msckf_inst<13, 512>(f, a);
    default: return launch_msckf_general(f, a);      // N > 208: any window length, everything in a global workspace
    }
#endif
}

template <int NT>
static int launch_usckf_inst(slk_filter *f, const KArgs &a)
{
    UCarve cv = carve_usckf(a.lay.N, a.lay.Nq, a.m, NT);
    size_t lds = (size_t)cv.total * sizeof(double);
    if (lds > 160 * 1024) { g_err = "state too large for the LDS-resident kernel"; return SLK_E_UNSUPPORTED; }
    auto kern = usckf_kernel<NT, 256>;
    int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), f->cfg.device, lds);
    if (rc_lds) return rc_lds;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds, f->stream, a);
    HIPCHECK(hipGetLastError());
    return SLK_OK;
}

// N <= 64, plain predict / update / step calls: three launches per step (predict and factorisation as one wave per filter,
// the update with the covariance left in global memory) -- the fused kernel above keeps the Tier-B modes and N > 64.
template <int NT>
static int launch_usckf_split(slk_filter *f, const KArgs &a0)
{
    KArgs a = a0;
    // The unit-test shape keeps the lower triangle (and the diagonal 16 x 16 tiles) of the covariance up to date only:
    // predict, the factorisation and the exact-shape update read nothing else (Usckf.hpp:537: Eigen::LLT); the strict upper
    // triangle is completed before anything else sees the matrix (mirror_upper).  -DSLK_USCKF_FULL_P: both triangles, always.
#ifndef SLK_USCKF_FULL_P
    if (NT == 3 && a.lay.N == 48 && a.lay.nfk == 3 && a.lay.nfkl == 9 && a.emit == 0 && !a.P_out && a.P == f->d_P
        && (!a.do_update || (a.m == 3 && a.mm == SLK_MM_VO_RELATIVE && a.gate <= 9))) {
        a.lower_only = 1;
        f->upper_stale = true;
    } else
#endif
    { int rcm = mirror_upper(f); if (rcm) return rcm; }
    if (a.do_predict) {
        hipLaunchKernelGGL(usckf_predict_kernel, dim3(a.B), dim3(64), 0, f->stream, a);
        HIPCHECK(hipGetLastError());
        if (!a.do_update) return SLK_OK;
        a.do_predict = 0;
    }
    // the exact shape with the plain update factors inside the update kernel (slk_usckf_fast.hpp): two launches per step,
    // no factor round trip through memory (-DSLK_USCKF_FACTOR_KERNEL: the three-launch form, for A/B runs)
    bool fused_factor = false;
#ifndef SLK_USCKF_FACTOR_KERNEL
    if constexpr (NT == 3)
        fused_factor = a.lay.nfk == 3 && a.lay.nfkl == 9 && a.m == 3 && a.emit == 0 && a.mm == SLK_MM_VO_RELATIVE && a.gate <= 9;
#endif
    int rc = SLK_OK;
    if (fused_factor) {
        a.wsL = nullptr;
        a.wsfail = nullptr;
    } else {
        rc = stage_reserve(f, f->ws_L, (size_t)a.B * pk_size(a.lay.N));
        if (rc) return rc;
        rc = stage_reserve(f, f->ws_DR, ((size_t)a.B * sizeof(int) + sizeof(double) - 1) / sizeof(double));
        if (rc) return rc;
        a.wsL = f->ws_L.p;
        a.wsfail = reinterpret_cast<int *>(f->ws_DR.p);
        if (NT == 3 && a.lay.N == 48) {             // the unit-test shape: the exact-size factor kernel (same N as 6 Msckf clones)
            hipLaunchKernelGGL((msckf_chol_kernel<3, 6>), dim3(a.B), dim3(64), 0, f->stream, a);
        } else {
            hipLaunchKernelGGL((msckf_chol_kernel<NT, -1>), dim3(a.B), dim3(64), 0, f->stream, a);
        }
        HIPCHECK(hipGetLastError());
    }
    UCarve cv = carve_usckf(a.lay.N, a.lay.Nq, a.m, NT, true);
    const size_t lds = (size_t)cv.total * sizeof(double);
#ifndef SLK_USCKF_UPD_THREADS
#define SLK_USCKF_UPD_THREADS 128   // two waves per filter: twice the filters in flight, fewer barrier waits (A/B: 256 -> 188 us, 128 -> 165 us, 64 -> 172 us at B = 4096)
#endif
    auto kern = usckf_kernel<NT, SLK_USCKF_UPD_THREADS, true>;
    size_t lds_fast = 0;
    if constexpr (NT == 3) {                    // the unit-test shape: exact instantiation (with its fast path's own LDS carve)
        if (a.lay.nfk == 3 && a.lay.nfkl == 9 && a.m == 3) { kern = usckf_kernel<3, SLK_USCKF_UPD_THREADS, true, true>; lds_fast = (size_t)UFast::total * sizeof(double); }
    }
    const size_t lds_use = lds > lds_fast ? lds : lds_fast;
    rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), f->cfg.device, lds_use);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(SLK_USCKF_UPD_THREADS), lds_use, f->stream, a);
    HIPCHECK(hipGetLastError());
    return SLK_OK;
}

static int launch_usckf(slk_filter *f, const KArgs &a)
{
    int NT = (a.lay.N + 15) / 16;
    if (f->upper_stale && (a.emit != 0 || a.lay.N > 48)) { int rcm = mirror_upper(f); if (rcm) return rcm; }   // (the fused kernels stage the whole matrix)
#if defined(SLK_DEV_N60) && defined(SLK_DEV_USCKF)       // (-DSLK_DEV_USCKF: the split path of N <= 48 only)
    if (NT == 3 && a.emit == 0) return launch_usckf_split<3>(f, a);
    g_err = "development build: Usckf split path of N <= 48 only"; return SLK_E_UNSUPPORTED;
#elif defined(SLK_DEV_N60)
    (void)NT; (void)f; (void)a;
    g_err = "development build: Msckf only"; return SLK_E_UNSUPPORTED;
#else
    const bool split = a.emit == 0 && a.lay.N <= 48;       // (usckf_predict_kernel stages 12 x N old rows and Fk in its 736 doubles of scratch)
    switch (NT) {
    case 3: return split ? launch_usckf_split<3>(f, a) : launch_usckf_inst<3>(f, a);
    case 4: return split ? launch_usckf_split<4>(f, a) : launch_usckf_inst<4>(f, a);
    case 5: return launch_usckf_inst<5>(f, a);
    case 6: return launch_usckf_inst<6>(f, a);
    default: g_err = "Usckf state dimension above 96 is not supported by this build"; return SLK_E_UNSUPPORTED;
    }
#endif
}

#ifdef SLK_STAMPS
static long long *g_dbg = nullptr;
static int g_stop = 0;
extern "C" void slk_debug_set_stamps(long long *device_buffer) { g_dbg = device_buffer; }   // [B][32], diagnostic build only
extern "C" void slk_debug_set_stop(int stamp) { g_stop = stamp; }                          // 0 = run to the end
#endif

static void base_args(slk_filter *f, KArgs &a)
{
    memset(&a, 0, sizeof(a));
#ifdef SLK_STAMPS
    a.dbg = g_dbg;
    a.stop = g_stop;
#endif
    a.B = f->B;
    a.lay = f->lay;
    a.mean = f->d_mean; a.P = f->d_P; a.status = f->d_status; a.outliers = f->d_outliers;
    a.rebuild_prec = f->rebuild_prec;
}

static int pm_inputs(int model) { return model == SLK_PM_CONST_VELOCITY ? 7 : ((model == SLK_PM_DELTA_POSE || model == SLK_PM_DEAD_RECKON) ? 13 : 0); }
static int mm_params(int model, int m)
{
    return model == SLK_MM_FEATURE_PROJ ? (m / 2) * 4 : (model == SLK_MM_POSE_POSITION ? 1 : 0);
}

static int fill_predict(slk_filter *f, KArgs &a, int model, const double *u, int u_stride,
                        const double *Q, int q_stride, int where)
{
    if (model != SLK_PM_CONST_VELOCITY && model != SLK_PM_DELTA_POSE && model != SLK_PM_DEAD_RECKON) return SLK_E_INVALID;
    if (!u || !Q) return SLK_E_INVALID;
    int nu = pm_inputs(model);
    if (u_stride != 0 && u_stride < nu) return SLK_E_INVALID;
    if (q_stride != 0 && q_stride < 144) return SLK_E_INVALID;
    a.do_predict = 1; a.pm = model; a.u_stride = u_stride; a.q_stride = q_stride;
    int rc = stage_in(f, f->st_u, u, u_stride ? (size_t)f->B * u_stride : (size_t)nu, where, &a.u);
    if (rc) return rc;
    return stage_in(f, f->st_Q, Q, q_stride ? (size_t)f->B * q_stride : (size_t)144, where, &a.Q);
}

static int fill_update(slk_filter *f, KArgs &a, int model, const double *params, int p_stride,
                       const double *z, int m, const double *R, int r_stride, int gate, int where)
{
    if (m < 1 || m > MAXM || !z || !R) return SLK_E_INVALID;
    if (r_stride != 0 && r_stride < m * m) return SLK_E_INVALID;
    if (f->lay.kind == SLK_MSCKF) {
        if (model != SLK_MM_FEATURE_PROJ && model != SLK_MM_POSE_POSITION && model != SLK_MODEL_EXTERNAL) return SLK_E_INVALID;
        if (model == SLK_MM_FEATURE_PROJ && (m & 1)) return SLK_E_INVALID;
        if (model == SLK_MM_POSE_POSITION && m != 3) return SLK_E_INVALID;
    } else {
        if (model != SLK_MM_VO_RELATIVE && model != SLK_MM_FEATURE_PROJ && model != SLK_MM_POSE_POSITION
            && model != SLK_MODEL_EXTERNAL) return SLK_E_INVALID;
        if (model == SLK_MM_VO_RELATIVE && (m != f->lay.nfk || m % 3)) return SLK_E_INVALID;
        if (model == SLK_MM_FEATURE_PROJ && (m & 1)) return SLK_E_INVALID;
        if (model == SLK_MM_POSE_POSITION && m != 3) return SLK_E_INVALID;
    }
    int np = mm_params(model, m);
    if (np && (!params || (p_stride != 0 && p_stride < np))) return SLK_E_INVALID;
    if (np && where == SLK_HOST) {
        // pose indices are caller data: reject anything outside 0..k (Msckf) / 0..2 (Usckf) before it reaches a kernel
        // (device-resident parameters are checked by the kernel itself: SLK_ST_BAD_INDEX)
        const double maxc = f->lay.kind == SLK_MSCKF ? (double)f->lay.k : 2.0;
        const int rows = p_stride ? f->B : 1;
        for (int b = 0; b < rows; ++b) {
            const double *row = params + (size_t)b * p_stride;
            if (model == SLK_MM_FEATURE_PROJ) {
                for (int q = 0; q < m / 2; ++q)
                    if (!(row[4 * q + 3] >= 0.0 && row[4 * q + 3] <= maxc)) { g_err = "pose index of a feature out of range"; return SLK_E_INVALID; }
            } else if (!(row[0] >= 0.0 && row[0] <= maxc)) { g_err = "pose index out of range"; return SLK_E_INVALID; }
        }
    }
    a.do_update = 1; a.mm = model; a.m = m; a.gate = gate; a.mp_stride = p_stride; a.r_stride = r_stride;
    int rc = np ? stage_in(f, f->st_mp, params, p_stride ? (size_t)f->B * p_stride : (size_t)np, where, &a.mp) : SLK_OK;
    if (rc) return rc;
    rc = stage_in(f, f->st_z, z, (size_t)f->B * m, where, &a.z);
    if (rc) return rc;
    return stage_in(f, f->st_R, R, r_stride ? (size_t)f->B * r_stride : (size_t)m * m, where, &a.R);
}

static int mirror_upper(slk_filter *f)
{
    if (!f->upper_stale) return SLK_OK;
    HIPCHECK(hipSetDevice(f->cfg.device));
    hipLaunchKernelGGL(slk_mirror_upper_kernel, dim3(f->B), dim3(256), 0, f->stream, f->d_P, f->lay.N);
    HIPCHECK(hipGetLastError());
    f->upper_stale = false;
    return SLK_OK;
}

static int launch(slk_filter *f, const KArgs &a)
{
    return f->lay.kind == SLK_MSCKF ? launch_msckf(f, a) : launch_usckf(f, a);
}

// Sliding window on the device: the reference leaves clone management to the caller
// (muState().sensorsk push/pop + setPk, Msckf.hpp:381-395; MultiState layout State.hpp:342, :373-396).
// op 1: append a clone of the current pose; its covariance rows / columns are those of the pose (J P J^T with
//       J = [I; E_pose], the MSCKF state augmentation for an identity sensor offset).
// op 2: drop clone `idx`: its 7 stored values and its 6 rows / columns disappear.
// grid (tiles of the new N x N matrix, B); the mean is moved by the first threads of tile 0.
__global__ void msckf_window_kernel(const double *mean, const double *P, double *nmean, double *nP, int k_old, int op, int idx)
{
    const int b = blockIdx.y;
    const int N = 12 + 6 * k_old, Nq = 13 + 7 * k_old;
    const int Nn = op == 1 ? N + 6 : N - 6, Nqn = op == 1 ? Nq + 7 : Nq - 7;
    const double *m = mean + (size_t)b * Nq, *Pb = P + (size_t)b * N * N;
    double *mo = nmean + (size_t)b * Nqn, *Po = nP + (size_t)b * Nn * Nn;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    auto tsrc = [&](int t) { return op == 1 ? (t < N ? t : t - N) : (t < 12 + 6 * idx ? t : t + 6); };
    if (e < Nn * Nn) {
        const int r = e % Nn, c = e / Nn;
        Po[e] = Pb[tsrc(r) + (size_t)tsrc(c) * N];
    }
    if (e < Nqn) {
        int ssrc;
        if (op == 1) ssrc = e < Nq ? e : e - Nq;             // pos[3] quat[4] of the current State sit at 0..6
        else ssrc = e < 13 + 7 * idx ? e : e + 7;
        mo[e] = m[ssrc];
    }
}

// checkSigmaPoints (Msckf.hpp:819-839), second half: compare the re-drawn mean / covariance with the filter's own.
// One workgroup per filter; res [2][B] = max |Pktest - Pk|, |mu_state [-] muX|.
__global__ void check_compare_kernel(Lay L, const double *mean, const double *P, const double *mean2, const double *P2, int B,
                                     double *res)
{
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x, N = L.N, Nq = L.Nq;
    const double *p = P + (size_t)b * N * N, *p2 = P2 + (size_t)b * N * N;
    const double *m = mean + (size_t)b * Nq, *m2 = mean2 + (size_t)b * Nq;
    double e = 0.0;
    for (int i = tid; i < N * N; i += 256) { double d = fabs(p2[i] - p[i]); e = (d > e || d != d) ? d : e; }
    red[tid] = e;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { double o = red[tid + s]; if (o > red[tid] || o != o) red[tid] = o; }
        __syncthreads();
    }
    const double cov_err = red[0];
    __syncthreads();
    double n2 = 0.0;
    for (int t = tid; t < N; t += 256) {
        int blk = 0, comp = 0, s = t2s(L, t, blk, comp);
        if (s >= 0) { double d = m[s] - m2[s]; n2 += d * d; }
        else if (comp == 0) {
            double dx, dy, dz;
            so3_boxminus(ldq(m + so3_soff(L, blk)), ldq(m2 + so3_soff(L, blk)), dx, dy, dz);
            n2 += dx * dx + dy * dy + dz * dz;
        }
    }
    red[tid] = n2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) { res[b] = cov_err; res[B + b] = sqrt(red[0]); }
}

// DeadReckon::updatePose delta poses of a batch (src/core/DeadReckon.hpp:129-239): one thread per filter
__global__ void dead_reckon_kernel(int B, const double *u, int u_stride, double *delta)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double uu[13], d[13];
    const double *src = u + (size_t)b * u_stride;
    for (int i = 0; i < 13; ++i) uu[i] = src[i];
    dead_reckon_delta(uu, d);
    for (int i = 0; i < 13; ++i) delta[(size_t)b * 13 + i] = d[i];
}

extern "C" {

int slk_dead_reckon(slk_filter *f, const double *u, int u_stride, double *delta, int where)
{
    if (!f || !u || !delta) return SLK_E_INVALID;
    if (u_stride != 0 && u_stride < 13) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    const double *du = nullptr;
    int rc = stage_in(f, f->st_u, u, u_stride ? (size_t)f->B * u_stride : (size_t)13, where, &du);
    if (rc) return rc;
    size_t n = (size_t)f->B * 13;
    double *dd = delta;
    if (where == SLK_HOST) { rc = stage_reserve(f, f->st_X, n); if (rc) return rc; dd = f->st_X.p; }
    hipLaunchKernelGGL(dead_reckon_kernel, dim3((f->B + 255) / 256), dim3(256), 0, f->stream, f->B, du, u_stride, dd);
    HIPCHECK(hipGetLastError());
    if (where == SLK_HOST) {
        HIPCHECK(hipMemcpyAsync(delta, dd, n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
        HIPCHECK(hipStreamSynchronize(f->stream));
    }
    return SLK_OK;
}

// stage an input of the pose ops through one of the handle's scratch buffers; out-of-place so that several inputs of
// one call do not share a buffer
static int stage_pose_in(slk_filter *f, Stage &s, const double *src, size_t n, int where, const double **out)
{
    return stage_in(f, s, src, n, where, out);
}

int slk_transform_compose(slk_filter *f, const double *t2, const double *cov2, const double *t1, const double *cov1,
                          double *t_out, double *cov_out, int additive, int where)
{
    if (!f || !t2 || !t1 || !t_out) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    const size_t B = (size_t)f->B;
    const double *d2, *d1, *dc2, *dc1;
    int rc = stage_pose_in(f, f->st_u, t2, B * 7, where, &d2);
    if (rc) return rc;
    rc = stage_pose_in(f, f->st_mp, t1, B * 7, where, &d1);
    if (rc) return rc;
    rc = stage_pose_in(f, f->st_X, cov2, B * 36, where, &dc2);
    if (rc) return rc;
    rc = stage_pose_in(f, f->st_Z, cov1, B * 36, where, &dc1);
    if (rc) return rc;
    double *dt = t_out, *dc = cov_out;
    if (where == SLK_HOST) {
        rc = stage_reserve(f, f->st_tmpM, B * 7);
        if (rc) return rc;
        rc = stage_reserve(f, f->st_tmpP, B * 36);
        if (rc) return rc;
        dt = f->st_tmpM.p;
        dc = cov_out ? f->st_tmpP.p : nullptr;
    }
    hipLaunchKernelGGL(transform_compose_kernel, dim3((f->B + 63) / 64), dim3(64), 0, f->stream, f->B, d2, dc2, d1, dc1, dt, dc,
                       additive);
    HIPCHECK(hipGetLastError());
    if (where == SLK_HOST) {
        HIPCHECK(hipMemcpyAsync(t_out, dt, B * 7 * sizeof(double), hipMemcpyDeviceToHost, f->stream));
        if (cov_out) HIPCHECK(hipMemcpyAsync(cov_out, dc, B * 36 * sizeof(double), hipMemcpyDeviceToHost, f->stream));
        HIPCHECK(hipStreamSynchronize(f->stream));
    }
    return SLK_OK;
}

int slk_dead_reckon_pose(slk_filter *f, const double *u, int u_stride, const double *velcov, int c_stride,
                         const double *prev, double *post, double *delta, int use_tf, int where)
{
    if (!f || !u || !velcov || !prev || !post) return SLK_E_INVALID;
    if ((u_stride != 0 && u_stride < 13) || (c_stride != 0 && c_stride < 36)) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    const size_t B = (size_t)f->B;
    const double *du, *dv, *dp;
    int rc = stage_pose_in(f, f->st_u, u, u_stride ? B * u_stride : 13, where, &du);
    if (rc) return rc;
    rc = stage_pose_in(f, f->st_Q, velcov, c_stride ? B * c_stride : 36, where, &dv);
    if (rc) return rc;
    rc = stage_pose_in(f, f->st_mp, prev, B * 25, where, &dp);
    if (rc) return rc;
    double *dpost = post, *ddelta = delta;
    if (where == SLK_HOST) {
        rc = stage_reserve(f, f->st_X, B * 49);
        if (rc) return rc;
        rc = stage_reserve(f, f->st_Z, B * 31);
        if (rc) return rc;
        dpost = f->st_X.p;
        ddelta = delta ? f->st_Z.p : nullptr;
        HIPCHECK(hipMemcpyAsync(dpost, post, B * 49 * sizeof(double), hipMemcpyHostToDevice, f->stream));
    }
    hipLaunchKernelGGL(dead_reckon_pose_kernel, dim3((f->B + 63) / 64), dim3(64), 0, f->stream, f->B, du, u_stride, dv, c_stride,
                       dp, dpost, ddelta, use_tf);
    HIPCHECK(hipGetLastError());
    if (where == SLK_HOST) {
        HIPCHECK(hipMemcpyAsync(post, dpost, B * 49 * sizeof(double), hipMemcpyDeviceToHost, f->stream));
        if (delta) HIPCHECK(hipMemcpyAsync(delta, ddelta, B * 31 * sizeof(double), hipMemcpyDeviceToHost, f->stream));
        HIPCHECK(hipStreamSynchronize(f->stream));
    }
    return SLK_OK;
}

int slk_predict(slk_filter *f, int model, const double *u, int u_stride, const double *Q, int q_stride, int where)
{
    if (!f) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    int rc = fill_predict(f, a, model, u, u_stride, Q, q_stride, where);
    if (rc) return rc;
    return launch(f, a);
}

int slk_update(slk_filter *f, int model, const double *params, int p_stride, const double *z, int m,
               const double *R, int r_stride, int gate, int where)
{
    if (!f || model == SLK_MODEL_EXTERNAL) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    int rc = fill_update(f, a, model, params, p_stride, z, m, R, r_stride, gate, where);
    if (rc) return rc;
    return launch(f, a);
}

int slk_update_ekf(slk_filter *f, const double *z, const double *zmean, const double *H, int m,
                   const double *R, int r_stride, int gate, int where)
{
    if (!f || f->lay.kind != SLK_MSCKF || !z || !zmean || !H || !R) return SLK_E_INVALID;
    const int N = f->lay.N;
    if (m < N || m > 512 || (m & 1)) return SLK_E_INVALID;       // reduceDimension needs m >= N rows; 2-row blocks
    if (r_stride != 0 && r_stride < m * m) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    { int rcm = mirror_upper(f); if (rcm) return rcm; }
    EkfArgs a;
    memset(&a, 0, sizeof(a));
    a.B = f->B; a.N = N; a.Nq = f->lay.Nq; a.k = f->lay.k; a.m = m; a.gate = gate;
    a.mean = f->d_mean; a.P = f->d_P; a.status = f->d_status; a.outliers = f->d_outliers;
    a.r_stride = r_stride;
    size_t B = (size_t)f->B;
    int rc = stage_in(f, f->st_z, z, B * m, where, &a.z);
    if (rc) return rc;
    rc = stage_in(f, f->st_mp, zmean, B * m, where, &a.zmean);
    if (rc) return rc;
    rc = stage_in(f, f->st_X, H, B * m * N, where, &a.H);
    if (rc) return rc;
    rc = stage_in(f, f->st_R, R, r_stride ? B * r_stride : (size_t)m * m, where, &a.R);
    if (rc) return rc;
    rc = stage_reserve(f, f->ws_ekf, B * ekf_ws_doubles(N, m));
    if (rc) return rc;
    a.ws = f->ws_ekf.p;
#ifdef SLK_STAMPS
    a.dbg = g_dbg;
#endif
#if defined(SLK_DEV_N60) && !defined(SLK_DEV_EKF)
    g_err = "development build: no EKF kernels"; return SLK_E_UNSUPPORTED;
#else
    const size_t lds = ekf_tile_lds_doubles(N, m) * sizeof(double);
    if (m <= 128 && N <= 64) {                                    // everything resident in LDS as 16 x 16 tiles
        auto kern = msckf_ekf_tile_kernel<1024>;
        rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), f->cfg.device, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(f->B), dim3(1024), lds, f->stream, a);
    } else {
#ifdef SLK_DEV_EKF
        g_err = "development build: EKF tile kernel only"; return SLK_E_UNSUPPORTED;
#else
        hipLaunchKernelGGL(msckf_ekf_kernel<256>, dim3(f->B), dim3(256), 0, f->stream, a);
#endif
    }
    HIPCHECK(hipGetLastError());
    return SLK_OK;
#endif
}

int slk_step(slk_filter *f, int pmodel, const double *u, int u_stride, const double *Q, int q_stride,
             int mmodel, const double *params, int p_stride, const double *z, int m,
             const double *R, int r_stride, int gate, int where)
{
    if (!f || mmodel == SLK_MODEL_EXTERNAL) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    int rc = fill_predict(f, a, pmodel, u, u_stride, Q, q_stride, where);
    if (rc) return rc;
    rc = fill_update(f, a, mmodel, params, p_stride, z, m, R, r_stride, gate, where);
    if (rc) return rc;
    return launch(f, a);
}

int slk_predict_sigma_points(slk_filter *f, double *X, int where)
{
    if (!f || !X) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    size_t n = (size_t)f->B * 25 * 13;
    a.emit = 1;
    if (where == SLK_DEVICE) a.Xout = X;
    else { int rc = stage_reserve(f, f->st_X, n); if (rc) return rc; a.Xout = f->st_X.p; }
    int rc = launch(f, a);
    if (rc) return rc;
    if (where == SLK_HOST) {
        HIPCHECK(hipMemcpyAsync(X, a.Xout, n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
        HIPCHECK(hipStreamSynchronize(f->stream));
    }
    return SLK_OK;
}

int slk_predict_from_sigma(slk_filter *f, const double *Y, const double *Q, int q_stride, int where)
{
    if (!f || !Y || !Q) return SLK_E_INVALID;
    if (q_stride != 0 && q_stride < 144) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    a.do_predict = 1; a.pm = SLK_MODEL_EXTERNAL; a.q_stride = q_stride;
    int rc = stage_in(f, f->st_X, Y, (size_t)f->B * 25 * 13, where, &a.Yext);
    if (rc) return rc;
    rc = stage_in(f, f->st_Q, Q, q_stride ? (size_t)f->B * q_stride : (size_t)144, where, &a.Q);
    if (rc) return rc;
    return launch(f, a);
}

int slk_update_sigma_points(slk_filter *f, double *X, int where)
{
    if (!f || !X) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    size_t n = (size_t)f->B * (2 * f->lay.N + 1) * f->lay.Nq;
    a.emit = 2; a.m = 1;
    if (where == SLK_DEVICE) a.Xout = X;
    else { int rc = stage_reserve(f, f->st_X, n); if (rc) return rc; a.Xout = f->st_X.p; }
    int rc = launch(f, a);
    if (rc) return rc;
    if (where == SLK_HOST) {
        HIPCHECK(hipMemcpyAsync(X, a.Xout, n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
        HIPCHECK(hipStreamSynchronize(f->stream));
    }
    return SLK_OK;
}

int slk_update_from_sigma(slk_filter *f, const double *Z, const double *z, int m, const double *R, int r_stride,
                          int gate, int where)
{
    if (!f || !Z) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    int rc = fill_update(f, a, SLK_MODEL_EXTERNAL, nullptr, 0, z, m, R, r_stride, gate, where);
    if (rc) return rc;
    rc = stage_in(f, f->st_Z, Z, (size_t)f->B * (2 * f->lay.N + 1) * m, where, &a.Zext);
    if (rc) return rc;
    return launch(f, a);
}

int slk_update_innovation(slk_filter *f, int model, const double *params, int p_stride, const double *Z,
                          const double *z, int m, const double *R, int r_stride, double *SI, int where)
{
    if (!f || !SI) return SLK_E_INVALID;
    if ((model == SLK_MODEL_EXTERNAL) != (Z != nullptr)) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    int rc = fill_update(f, a, model, params, p_stride, z, m, R, r_stride, 0, where);
    if (rc) return rc;
    if (Z) { rc = stage_in(f, f->st_Z, Z, (size_t)f->B * (2 * f->lay.N + 1) * m, where, &a.Zext); if (rc) return rc; }
    const size_t n = (size_t)f->B * (m * m + m);
    a.emit = 4;
    if (where == SLK_DEVICE) a.Xout = SI;
    else { rc = stage_reserve(f, f->st_X, n); if (rc) return rc; a.Xout = f->st_X.p; }
    rc = launch(f, a);
    if (rc) return rc;
    if (where == SLK_HOST) {
        HIPCHECK(hipMemcpyAsync(SI, a.Xout, n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
        HIPCHECK(hipStreamSynchronize(f->stream));
    }
    return SLK_OK;
}

int slk_update_selected(slk_filter *f, int model, const double *params, int p_stride, const double *Z,
                        const double *z, int m, const double *R, int r_stride, const int *rowsel, int where)
{
    if (!f || f->lay.kind != SLK_MSCKF || !rowsel) return SLK_E_INVALID;
    if ((model == SLK_MODEL_EXTERNAL) != (Z != nullptr)) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    KArgs a;
    base_args(f, a);
    int rc = fill_update(f, a, model, params, p_stride, z, m, R, r_stride, 2, where);
    if (rc) return rc;
    if (Z) { rc = stage_in(f, f->st_Z, Z, (size_t)f->B * (2 * f->lay.N + 1) * m, where, &a.Zext); if (rc) return rc; }
    const size_t n = (size_t)f->B * (m + 2);
    if (where == SLK_DEVICE) {
        a.rowsel = rowsel;
    } else {
        for (size_t b = 0; b < (size_t)f->B; ++b) {                 // host data: validate before it reaches the kernel
            const int *rs = rowsel + b * (m + 2);
            if (rs[0] < 0 || rs[0] > m || rs[1] < 0) return SLK_E_INVALID;
            for (int r = 0; r < rs[0]; ++r) if (rs[2 + r] < 0 || rs[2 + r] >= m) return SLK_E_INVALID;
        }
        rc = stage_reserve(f, f->st_tmpM, (n * sizeof(int) + sizeof(double) - 1) / sizeof(double));
        if (rc) return rc;
        HIPCHECK(hipMemcpyAsync(f->st_tmpM.p, rowsel, n * sizeof(int), hipMemcpyHostToDevice, f->stream));
        a.rowsel = reinterpret_cast<const int *>(f->st_tmpM.p);
    }
    return launch(f, a);
}

int slk_get_outliers(slk_filter *f, unsigned *outliers, int where)
{
    if (!f || !outliers) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    HIPCHECK(hipMemcpyAsync(outliers, f->d_outliers, (size_t)f->B * sizeof(unsigned),
                            where == SLK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, f->stream));
    if (where == SLK_HOST) HIPCHECK(hipStreamSynchronize(f->stream));
    return SLK_OK;
}

int slk_get_status(slk_filter *f, int *status, int where)
{
    if (!f || !status) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    HIPCHECK(hipMemcpyAsync(status, f->d_status, (size_t)f->B * sizeof(int),
                            where == SLK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, f->stream));
    if (where == SLK_HOST) HIPCHECK(hipStreamSynchronize(f->stream));
    return SLK_OK;
}

int slk_clear_status(slk_filter *f)
{
    if (!f) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    HIPCHECK(hipMemsetAsync(f->d_status, 0, (size_t)f->B * sizeof(int), f->stream));
    return SLK_OK;
}

int slk_set_rebuild_precision(slk_filter *f, int mode)
{
    if (!f || mode < SLK_PREC_F64 || mode > SLK_PREC_BF16) return SLK_E_INVALID;
    f->rebuild_prec = mode;
    return SLK_OK;
}

int slk_sync(slk_filter *f)
{
    if (!f) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    HIPCHECK(hipStreamSynchronize(f->stream));
    return SLK_OK;
}

int slk_timer_start(slk_filter *f)
{
    if (!f) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    HIPCHECK(hipEventRecord(f->ev0, f->stream));
    return SLK_OK;
}

int slk_timer_stop(slk_filter *f, float *ms)
{
    if (!f || !ms) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    HIPCHECK(hipEventRecord(f->ev1, f->stream));
    HIPCHECK(hipEventSynchronize(f->ev1));
    HIPCHECK(hipEventElapsedTime(ms, f->ev0, f->ev1));
    return SLK_OK;
}

// ---- Usckf bookkeeping (Usckf.hpp:322-433): block copies on the device
int slk_usckf_cloning(slk_filter *f, int mode)
{
    if (!f || f->lay.kind != SLK_USCKF) return SLK_E_INVALID;
    if (mode != SLK_STATEK_I && mode != SLK_STATEK_L) return SLK_OK;   // default: break (Usckf.hpp:428-429)
    HIPCHECK(hipSetDevice(f->cfg.device));
    { int rcm = mirror_upper(f); if (rcm) return rcm; }          // (whole blocks of P are copied)
    int total = f->B * 256;
    hipLaunchKernelGGL(usckf_cloning_kernel, dim3((total + 255) / 256), dim3(256), 0, f->stream,
                       f->d_mean, f->d_P, f->B, f->lay.N, f->lay.Nq, mode);
    HIPCHECK(hipGetLastError());
    return SLK_OK;
}

// The second pair of state buffers (layout changes are built into it, then the pairs swap): make it hold at least
// need_mean / need_P doubles.  Grows with headroom so that a sliding window does not come back here; a hipMalloc
// happens only then.  Nothing of the handle is changed on failure.
static int reserve_alt(slk_filter *f, size_t need_mean, size_t need_P)
{
    if (f->cap_mean_alt < need_mean) {
        if (f->d_mean_alt) HIPCHECK(hipFree(f->d_mean_alt));        // (hipFree waits for work that still reads it)
        f->d_mean_alt = nullptr; f->cap_mean_alt = 0;
        const size_t want = need_mean + need_mean / 2;
        HIPCHECK(hipMalloc(&f->d_mean_alt, want * sizeof(double)));
        f->cap_mean_alt = want;
    }
    if (f->cap_P_alt < need_P) {
        if (f->d_P_alt) HIPCHECK(hipFree(f->d_P_alt));
        f->d_P_alt = nullptr; f->cap_P_alt = 0;
        const size_t want = need_P + need_P / 2;
        HIPCHECK(hipMalloc(&f->d_P_alt, want * sizeof(double)));
        f->cap_P_alt = want;
    }
    return SLK_OK;
}

static void swap_state_buffers(slk_filter *f)
{
    std::swap(f->d_mean, f->d_mean_alt);
    std::swap(f->d_P, f->d_P_alt);
    std::swap(f->cap_mean, f->cap_mean_alt);
    std::swap(f->cap_P, f->cap_P_alt);
}

int slk_usckf_set_measurement(slk_filter *f, int mode, const double *z, int n, const double *R, int where)
{
    if (!f || f->lay.kind != SLK_USCKF || !z || !R || n < 1) return SLK_E_INVALID;
    if (mode != SLK_STATEK && mode != SLK_STATEK_L) return SLK_OK;
    HIPCHECK(hipSetDevice(f->cfg.device));
    { int rcm = mirror_upper(f); if (rcm) return rcm; }          // (whole blocks of P are copied)
    Lay oldL = f->lay;
    int nfk = mode == SLK_STATEK ? n : oldL.nfk, nfkl = mode == SLK_STATEK_L ? n : oldL.nfkl;
    Lay newL = make_lay(SLK_USCKF, 0, nfk, nfkl);
    if (newL.N > 96) { g_err = "Usckf state dimension above 96 is not supported by this build"; return SLK_E_UNSUPPORTED; }
    size_t B = (size_t)f->B;
    const double *dz, *dR;
    int rc = stage_in(f, f->st_z, z, B * n, where, &dz);
    if (rc) return rc;
    rc = stage_in(f, f->st_R, R, (size_t)n * n, where, &dR);
    if (rc) return rc;
    // built out of place into the second buffer pair on the stream, then the pairs swap: no allocation in the steady
    // state, no host synchronisation
    rc = reserve_alt(f, B * newL.Nq, B * (size_t)newL.N * newL.N);
    if (rc) return rc;
    int total = f->B * newL.N * newL.N;
    hipLaunchKernelGGL(usckf_set_measurement_kernel, dim3((total + 255) / 256), dim3(256), 0, f->stream,
                       f->d_mean, f->d_P, f->d_mean_alt, f->d_P_alt, dz, dR, f->B, oldL.nfk, oldL.nfkl, nfk, nfkl, mode, n);
    HIPCHECK(hipGetLastError());
    swap_state_buffers(f);
    f->lay = newL;
    f->cfg.n_featuresk = nfk; f->cfg.n_featuresk_l = nfkl;
    return SLK_OK;
}

static int msckf_window_op(slk_filter *f, int op, int idx)
{
    if (!f || f->lay.kind != SLK_MSCKF) return SLK_E_INVALID;
    const int k_old = f->lay.k, k_new = op == 1 ? k_old + 1 : k_old - 1;
    if (op == 2 && (idx < 0 || idx >= k_old)) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    { int rcm = mirror_upper(f); if (rcm) return rcm; }          // (the window kernel copies whole blocks of P)
    Lay newL = make_lay(SLK_MSCKF, k_new, 0, 0);
    size_t B = (size_t)f->B;
    // push / pop on the stream into the second buffer pair (no hipMalloc, no synchronisation once it has its size)
    int rc = reserve_alt(f, B * newL.Nq, B * (size_t)newL.N * newL.N);
    if (rc) return rc;
    hipLaunchKernelGGL(msckf_window_kernel, dim3((newL.N * newL.N + 255) / 256, f->B), dim3(256), 0, f->stream,
                       f->d_mean, f->d_P, f->d_mean_alt, f->d_P_alt, k_old, op, idx);
    HIPCHECK(hipGetLastError());
    swap_state_buffers(f);
    f->lay = newL;
    f->cfg.n_clones = k_new;
    return SLK_OK;
}

int slk_msckf_clone_pose(slk_filter *f) { return msckf_window_op(f, 1, 0); }
int slk_msckf_drop_clone(slk_filter *f, int index) { return msckf_window_op(f, 2, index); }

int slk_msckf_resize(slk_filter *f, int n_clones)
{
    if (!f || f->lay.kind != SLK_MSCKF || n_clones < 0) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    f->upper_stale = false;                                        // (the state is zeroed below)
    Lay newL = make_lay(SLK_MSCKF, n_clones, 0, 0);
    size_t B = (size_t)f->B;
    const size_t need_mean = B * newL.Nq, need_P = B * (size_t)newL.N * newL.N;
    if (need_mean > f->cap_mean || need_P > f->cap_P) {            // the zeroed state goes to the second pair, then swap
        int rc = reserve_alt(f, need_mean, need_P);
        if (rc) return rc;
        swap_state_buffers(f);
    }
    HIPCHECK(hipMemsetAsync(f->d_mean, 0, need_mean * sizeof(double), f->stream));
    HIPCHECK(hipMemsetAsync(f->d_P, 0, need_P * sizeof(double), f->stream));
    f->lay = newL;
    f->cfg.n_clones = n_clones;
    return SLK_OK;
}

int slk_check_sigma_points(slk_filter *f, double *max_cov_err, double *mean_err, int where)
{
    if (!f || f->lay.kind != SLK_MSCKF || !max_cov_err || !mean_err) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(f->cfg.device));
    const size_t B = (size_t)f->B, N = (size_t)f->lay.N, Nq = (size_t)f->lay.Nq;
    int rc = stage_reserve(f, f->st_tmpP, B * N * N);
    if (rc) return rc;
    rc = stage_reserve(f, f->st_tmpM, B * Nq + 2 * B);
    if (rc) return rc;
    KArgs a;
    base_args(f, a);
    a.emit = 3; a.m = 1;
    a.P_out = f->st_tmpP.p;
    a.mean_out = f->st_tmpM.p;
    rc = launch(f, a);
    if (rc) return rc;
    double *res = f->st_tmpM.p + B * Nq;                   // [2][B]
    hipLaunchKernelGGL(check_compare_kernel, dim3(f->B), dim3(256), 0, f->stream, f->lay, (const double *)f->d_mean,
                       (const double *)f->d_P, (const double *)a.mean_out, (const double *)a.P_out, f->B, res);
    HIPCHECK(hipGetLastError());
    const hipMemcpyKind kind = where == SLK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    HIPCHECK(hipMemcpyAsync(max_cov_err, res, B * sizeof(double), kind, f->stream));
    HIPCHECK(hipMemcpyAsync(mean_err, res + B, B * sizeof(double), kind, f->stream));
    if (where == SLK_HOST) HIPCHECK(hipStreamSynchronize(f->stream));
    return SLK_OK;
}

struct slk_adaptive {
    int B, device;
    unsigned m1, m2, r1count;
    double gamma;
    hipStream_t stream;
    bool own_stream;
    double *d_hist = nullptr;
    unsigned *d_r2 = nullptr;
    Stage st[6];            // host-input staging: xk, Pk, z, H, R, Rout
};

static int adaptive_stage(slk_adaptive *a, Stage &s, const double *src, size_t n, int where, const double **out)
{
    if (where == SLK_DEVICE) { *out = src; return SLK_OK; }
    if (s.cap < n) {
        if (s.p) HIPCHECK(hipFree(s.p));
        s.p = nullptr; s.cap = 0;
        HIPCHECK(hipMalloc(&s.p, n * sizeof(double)));
        s.cap = n;
    }
    HIPCHECK(hipMemcpyAsync(s.p, src, n * sizeof(double), hipMemcpyHostToDevice, a->stream));
    *out = s.p;
    return SLK_OK;
}

int slk_adaptive_create(int batch, int device, unsigned m1, unsigned m2, double gamma, unsigned r2count, void *stream,
                        slk_adaptive **out)
{
    if (!out || batch < 1 || m1 < 1) return SLK_E_INVALID;
    int ndev = slk_device_count();
    if (ndev <= 0) { g_err = "no HIP device: the slk library has no CPU fallback"; return SLK_E_NO_DEVICE; }
    if (device < 0 || device >= ndev) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(device));
    slk_adaptive *a = new slk_adaptive();
    a->B = batch; a->device = device; a->m1 = m1; a->m2 = m2; a->gamma = gamma; a->r1count = 0;   // :158-160
    a->own_stream = stream == nullptr;
    if (a->own_stream) {
        if (hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) != hipSuccess) { delete a; return SLK_E_HIP; }
    } else a->stream = (hipStream_t)stream;
    std::vector<unsigned> r2((size_t)batch, r2count);
    bool ok = hipMalloc(&a->d_hist, (size_t)batch * m1 * 9 * sizeof(double)) == hipSuccess
           && hipMalloc(&a->d_r2, (size_t)batch * sizeof(unsigned)) == hipSuccess
           && hipMemsetAsync(a->d_hist, 0, (size_t)batch * m1 * 9 * sizeof(double), a->stream) == hipSuccess      // :162-165
           && hipMemcpyAsync(a->d_r2, r2.data(), (size_t)batch * sizeof(unsigned), hipMemcpyHostToDevice, a->stream) == hipSuccess
           && hipStreamSynchronize(a->stream) == hipSuccess;
    if (!ok) { g_err = "device allocation failed"; slk_adaptive_destroy(a); return SLK_E_NOMEM; }
    *out = a;
    return SLK_OK;
}

void slk_adaptive_destroy(slk_adaptive *a)
{
    if (!a) return;
    (void)hipSetDevice(a->device);
    (void)hipStreamSynchronize(a->stream);
    for (Stage &s : a->st) if (s.p) (void)hipFree(s.p);
    if (a->d_hist) (void)hipFree(a->d_hist);
    if (a->d_r2) (void)hipFree(a->d_r2);
    if (a->own_stream) (void)hipStreamDestroy(a->stream);
    delete a;
}

int slk_adaptive_matrix(slk_adaptive *a, int n, const double *xk, const double *Pk, const double *z, const double *H,
                        const double *R, int r_stride, double *Rout, int where)
{
    if (!a || n < 1 || !xk || !Pk || !z || !H || !R || !Rout) return SLK_E_INVALID;
    if (r_stride != 0 && r_stride < 9) return SLK_E_INVALID;
    HIPCHECK(hipSetDevice(a->device));
    const size_t B = (size_t)a->B;
    const double *dx, *dP, *dz, *dH, *dR;
    int rc = adaptive_stage(a, a->st[0], xk, B * n, where, &dx); if (rc) return rc;
    rc = adaptive_stage(a, a->st[1], Pk, B * n * n, where, &dP); if (rc) return rc;
    rc = adaptive_stage(a, a->st[2], z, B * 3, where, &dz); if (rc) return rc;
    rc = adaptive_stage(a, a->st[3], H, B * 3 * n, where, &dH); if (rc) return rc;
    rc = adaptive_stage(a, a->st[4], R, r_stride ? B * r_stride : 9, where, &dR); if (rc) return rc;
    double *dout = Rout;
    if (where == SLK_HOST) {
        Stage &s = a->st[5];
        if (s.cap < B * 9) {
            if (s.p) HIPCHECK(hipFree(s.p));
            s.p = nullptr; s.cap = 0;
            HIPCHECK(hipMalloc(&s.p, B * 9 * sizeof(double)));
            s.cap = B * 9;
        }
        dout = s.p;
    }
    hipLaunchKernelGGL(adaptive_attitude_cov_kernel, dim3((a->B + 127) / 128), dim3(128), 0, a->stream, a->B, a->m1, a->m2, a->gamma,
                       a->d_hist, a->r1count, a->d_r2, n, dx, dP, dz, dH, dR, r_stride, dout);
    HIPCHECK(hipGetLastError());
    a->r1count = (a->r1count + 1) % a->m1;                                           // :213
    if (where == SLK_HOST) {
        HIPCHECK(hipMemcpyAsync(Rout, dout, B * 9 * sizeof(double), hipMemcpyDeviceToHost, a->stream));
        HIPCHECK(hipStreamSynchronize(a->stream));
    }
    return SLK_OK;
}

int slk_selftest_mfma(int device)
{
    if (slk_device_count() <= 0) { g_err = "no HIP device"; return SLK_E_NO_DEVICE; }
    HIPCHECK(hipSetDevice(device));
    double hA[64], hB[64], hC[256], *dA, *dB, *dC;
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.37 * i - 0.01 * i * i; hB[i] = -2.0 + 0.11 * i + 0.003 * i * i; }
    HIPCHECK(hipMalloc(&dA, sizeof(hA)));
    HIPCHECK(hipMalloc(&dB, sizeof(hB)));
    HIPCHECK(hipMalloc(&dC, sizeof(hC)));
    HIPCHECK(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    int bad = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            double ref = 0;
            for (int k = 0; k < 4; ++k) ref += hA[r * 4 + k] * hB[k * 16 + c];
            double err = hC[r * 16 + c] - ref;
            if (err < 0) err = -err;
            if (err > 1e-9) ++bad;
        }
    return bad;
}

} // extern "C"
