// slk_ekf.hpp -- Msckf EKF update on the GPU (reference src/filters/Msckf.hpp:284-349).
//
// zmean = h(mu) and the Jacobian H (m x N) come from the caller's functor (:310).  One workgroup per filter; the
// dense work runs on a per-filter global workspace (it stays in L2 / Infinity Cache): the straightforward fp64
// restatement of the reference's steps, kept for the shapes the LDS-resident tile kernel (slk_ekf_tiles.hpp: m <= 128,
// N <= 64, everything on the matrix cores) does not hold.
//
//   removeOutliers (:756-789): information = (H P H^T + R)^-1 ONCE (Cholesky here: S0 is SPD for a valid R), its
//     2x2 diagonal blocks indexed with the RUNNING block number while rows are erased with the reference's shifted
//     second erase;  reduceDimension (:791-816): Householder QR with Eigen's reflector convention
//     (makeHouseholder: beta = -sign(c0)||x||, tau = (beta - c0)/beta, tau = 0 for an exactly zero tail),
//     thinQ = Q * I(m', N), H <- R(0:N, 0:N), innovation <- thinQ^T innovation, R <- thinQ^T R thinQ;
//   S = H P H^T + R, K = P H^T S^-1 (:324-325), Pk -= K S K^T (:330), mu <- mu [+] K innovation (:331);
//   base::guaranteeSPD's result is discarded by the reference (:340): nothing to do.
#pragma once
#include "slk_kernels.hpp"

namespace slk {

struct EkfArgs {
    int B, N, Nq, k, m, gate;
    double *mean, *P;
    int *status;
    unsigned *outliers;
    const double *z, *zmean, *H, *R;     // [B][m], [B][m], [B][m*N] column-major, [B or 1][m*m]
    int r_stride;
    double *ws;                          // per-filter workspace, ekf_ws_doubles(N, m) each
    long long *dbg;                      // phase stamps [B][32], diagnostic builds only
};

__host__ __device__ inline size_t ekf_ws_doubles(int N, int m)
{
    return (size_t)N * m + 3 * (size_t)m * m + 3 * (size_t)m * N + 6 * (size_t)N * N + 4 * (size_t)m + 4 * (size_t)N + 64;
}

#define EKF_AT(M, ld, i, j) (M)[(size_t)(j) * (ld) + (i)]
#ifdef SLK_STAMPS
#define EKF_STAMP(i) do { if (threadIdx.x == 0 && a.dbg) a.dbg[(size_t)blockIdx.x * 32 + (i)] = clock64(); } while (0)
#else
#define EKF_STAMP(i) do { } while (0)
#endif

// lower Cholesky in place (column by column, right-looking); returns through *flag the first non-positive pivot
template <int NTHREADS>
__device__ void ekf_cholesky(double *A, int n, int tid, int *flag)
{
    for (int j = 0; j < n; ++j) {
        const double d = EKF_AT(A, n, j, j);
        if (!(d > 0.0)) { if (tid == 0 && *flag < 0) *flag = j; }
        __syncthreads();
        const double s = sqrt(d);
        for (int i = j + 1 + tid; i < n; i += NTHREADS) EKF_AT(A, n, i, j) /= s;
        if (tid == 0) EKF_AT(A, n, j, j) = s;
        __syncthreads();
        // trailing update of the lower triangle: A[i, c] -= L[i, j] L[c, j], j < c <= i
        const int rem = n - j - 1;
        for (int e = tid; e < rem * rem; e += NTHREADS) {
            const int c = j + 1 + e / rem, i = j + 1 + e % rem;
            if (i >= c) EKF_AT(A, n, i, c) -= EKF_AT(A, n, i, j) * EKF_AT(A, n, c, j);
        }
        __syncthreads();
    }
}

template <int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void msckf_ekf_kernel(EkfArgs a)
{
    __shared__ int idx[520];
    __shared__ int sh[8];                 // 0 count, 1 outliers, 2 flag S0, 3 flag S
    __shared__ double red[NTHREADS];
    __shared__ double hh[4];              // beta, tau of the current reflector
    const int tid = threadIdx.x, b = blockIdx.x;
    const int N = a.N, Nq = a.Nq, m = a.m;
    double *mean = a.mean + (size_t)b * Nq, *P = a.P + (size_t)b * N * N;
    const double *z = a.z + (size_t)b * m, *zm = a.zmean + (size_t)b * m, *H = a.H + (size_t)b * m * N;
    const double *R = a.R + (size_t)b * a.r_stride;
    double *w = a.ws + (size_t)b * ekf_ws_doubles(N, m);
    double *PHt = w;                w += (size_t)N * m;      // N x m
    double *S0 = w;                 w += (size_t)m * m;      // m x m -> its Cholesky factor
    double *Li = w;                 w += (size_t)m * m;      // inverse of that factor
    double *Rr = w;                 w += (size_t)m * m;      // gated R (m' x m')
    double *Hq = w;                 w += (size_t)m * N;      // gated H (m' x N) -> QR in place
    double *Q1 = w;                 w += (size_t)m * N;      // thinQ
    double *T1 = w;                 w += (size_t)m * N;      // Rr * thinQ
    double *Hr = w;                 w += (size_t)N * N;
    double *Rn = w;                 w += (size_t)N * N;
    double *T2 = w;                 w += (size_t)N * N;      // P Hr^T
    double *S = w;                  w += (size_t)N * N;      // -> Cholesky factor
    double *K = w;                  w += (size_t)N * N;
    double *Pn = w;                 w += (size_t)N * N;
    double *innov = w;              w += m;
    double *rq = w;                 w += m;
    double *tau = w;                w += 2 * m;
    double *rn = w;                 w += N;
    double *delta = w;              w += 3 * N;
    if (tid == 0) { sh[0] = m; sh[1] = 0; sh[2] = -1; sh[3] = -1; a.outliers[b] = 0u; }
    for (int r = tid; r < m; r += NTHREADS) { innov[r] = z[r] - zm[r]; idx[r] = r; }              // :312
    // P H^T and S0 = H P H^T + R (:765-766)
    for (int e = tid; e < N * m; e += NTHREADS) {
        const int i = e % N, j = e / N;
        double s = 0.0;
        for (int p = 0; p < N; ++p) s += EKF_AT(P, N, i, p) * EKF_AT(H, m, j, p);
        EKF_AT(PHt, N, i, j) = s;
    }
    __syncthreads();
    for (int e = tid; e < m * m; e += NTHREADS) {
        const int i = e % m, j = e / m;
        double s = 0.0;
        for (int p = 0; p < N; ++p) s += EKF_AT(H, m, i, p) * EKF_AT(PHt, N, p, j);
        EKF_AT(S0, m, i, j) = s + EKF_AT(R, m, i, j);
    }
    __syncthreads();
    ekf_cholesky<NTHREADS>(S0, m, tid, &sh[2]);
    int status = 0;
    if (sh[2] >= 0) {
        status |= SLK_ST_SINGULAR;              // the reference would invert an indefinite matrix with PartialPivLU
    } else {
        // Li = L^-1 (lower), one column per thread
        for (int c = tid; c < m; c += NTHREADS) {
            for (int i = 0; i < m; ++i) {
                double s = (i == c) ? 1.0 : 0.0;
                if (i < c) { EKF_AT(Li, m, i, c) = 0.0; continue; }
                for (int p = c; p < i; ++p) s -= EKF_AT(S0, m, i, p) * EKF_AT(Li, m, p, c);
                EKF_AT(Li, m, i, c) = s / EKF_AT(S0, m, i, i);
            }
        }
        __syncthreads();
        // removeOutliers (:767-787); information(a, b) = sum_k Li(k, a) Li(k, b)
        if (tid == 0) {
            int cnt = m;
            unsigned nout = 0;
            int i = 0;
            auto info = [&](int r, int c) {
                double s = 0.0;
                for (int p = (r > c ? r : c); p < m; ++p) s += EKF_AT(Li, m, p, r) * EKF_AT(Li, m, p, c);
                return s;
            };
            while (i < cnt / 2) {
                const double r0 = innov[idx[2 * i]], r1 = innov[idx[2 * i + 1]];
                const int ia = 2 * i, ib = 2 * i + 1;                  // block of the UNREDUCED information matrix
                const double i00 = info(ia, ia), i01 = info(ia, ib), i11 = info(ib, ib);
                const double d2 = r0 * (i00 * r0 + i01 * r1) + r1 * (i01 * r0 + i11 * r1);
                const bool ok = a.gate ? (d2 < 5.99) : true;           // chi2_0.95(2), Msckf.hpp:861-865
                if (!ok) {
                    for (int rep = 0; rep < 2; ++rep) {                // removeRow semantics, :688-697
                        int pos = 2 * i + rep, numRows = cnt - 1;
                        if (pos < numRows) for (int q = pos; q < numRows; ++q) idx[q] = idx[q + 1];
                        cnt = numRows;
                    }
                    nout++;
                } else {
                    i++;
                }
            }
            sh[0] = cnt;
            sh[1] = (int)nout;
            a.outliers[b] = nout;
        }
        __syncthreads();
        const int mm = sh[0];
        if (mm > 0 && mm < N) {
            status |= SLK_ST_EKF_ROWS;          // reduceDimension would read R.block(0,0,N,N) out of range (:806)
        } else if (mm > 0) {
            // gated copies
            for (int e = tid; e < mm * N; e += NTHREADS) { const int i = e % mm, j = e / mm; EKF_AT(Hq, mm, i, j) = EKF_AT(H, m, idx[i], j); }
            for (int e = tid; e < mm * mm; e += NTHREADS) { const int i = e % mm, j = e / mm; EKF_AT(Rr, mm, i, j) = EKF_AT(R, m, idx[i], idx[j]); }
            for (int i = tid; i < mm; i += NTHREADS) rq[i] = innov[idx[i]];
            __syncthreads();
            // Householder QR of Hq (mm x N), :797
            for (int kk = 0; kk < N; ++kk) {
                double part = 0.0;
                for (int i = kk + 1 + tid; i < mm; i += NTHREADS) { const double v = EKF_AT(Hq, mm, i, kk); part += v * v; }
                red[tid] = part;
                __syncthreads();
                if (tid == 0) {
                    double tail = 0.0;
                    for (int t = 0; t < NTHREADS; ++t) tail += red[t];
                    const double c0 = EKF_AT(Hq, mm, kk, kk);
                    double beta, tk, den = 1.0;
                    if (tail <= 2.2250738585072014e-308) { tk = 0.0; beta = c0; den = 0.0; }
                    else {
                        beta = sqrt(c0 * c0 + tail);
                        if (c0 >= 0.0) beta = -beta;
                        den = c0 - beta;
                        tk = (beta - c0) / beta;
                    }
                    hh[0] = beta; hh[1] = tk; hh[2] = den;
                    tau[kk] = tk;
                }
                __syncthreads();
                const double tk = hh[1], den = hh[2];
                for (int i = kk + 1 + tid; i < mm; i += NTHREADS) EKF_AT(Hq, mm, i, kk) = (den != 0.0) ? EKF_AT(Hq, mm, i, kk) / den : 0.0;
                if (tid == 0) EKF_AT(Hq, mm, kk, kk) = hh[0];
                __syncthreads();
                for (int j = kk + 1 + tid; j < N; j += NTHREADS) {          // one trailing column per thread
                    double wv = EKF_AT(Hq, mm, kk, j);
                    for (int i = kk + 1; i < mm; ++i) wv += EKF_AT(Hq, mm, i, kk) * EKF_AT(Hq, mm, i, j);
                    wv *= tk;
                    EKF_AT(Hq, mm, kk, j) -= wv;
                    for (int i = kk + 1; i < mm; ++i) EKF_AT(Hq, mm, i, j) -= EKF_AT(Hq, mm, i, kk) * wv;
                }
                __syncthreads();
            }
            // thinQ = Q * I(mm, N) (:802-803): one column per thread, reflectors applied in reverse
            for (int j = tid; j < N; j += NTHREADS) {
                for (int i = 0; i < mm; ++i) EKF_AT(Q1, mm, i, j) = (i == j) ? 1.0 : 0.0;
                for (int kk = N - 1; kk >= 0; --kk) {
                    double wv = EKF_AT(Q1, mm, kk, j);
                    for (int i = kk + 1; i < mm; ++i) wv += EKF_AT(Hq, mm, i, kk) * EKF_AT(Q1, mm, i, j);
                    wv *= tau[kk];
                    EKF_AT(Q1, mm, kk, j) -= wv;
                    for (int i = kk + 1; i < mm; ++i) EKF_AT(Q1, mm, i, j) -= EKF_AT(Hq, mm, i, kk) * wv;
                }
            }
            __syncthreads();
            // reduced quantities (:806-812)
            for (int e = tid; e < N * N; e += NTHREADS) { const int i = e % N, j = e / N; EKF_AT(Hr, N, i, j) = (i <= j) ? EKF_AT(Hq, mm, i, j) : 0.0; }
            for (int j = tid; j < N; j += NTHREADS) {
                double s = 0.0;
                for (int i = 0; i < mm; ++i) s += EKF_AT(Q1, mm, i, j) * rq[i];
                rn[j] = s;
            }
            for (int e = tid; e < mm * N; e += NTHREADS) {
                const int i = e % mm, j = e / mm;
                double s = 0.0;
                for (int p = 0; p < mm; ++p) s += EKF_AT(Rr, mm, i, p) * EKF_AT(Q1, mm, p, j);
                EKF_AT(T1, mm, i, j) = s;
            }
            __syncthreads();
            for (int e = tid; e < N * N; e += NTHREADS) {
                const int i = e % N, j = e / N;
                double s = 0.0;
                for (int p = 0; p < mm; ++p) s += EKF_AT(Q1, mm, p, i) * EKF_AT(T1, mm, p, j);
                EKF_AT(Rn, N, i, j) = s;
                double t = 0.0;                                            // T2 = P Hr^T
                for (int p = j; p < N; ++p) t += EKF_AT(P, N, i, p) * EKF_AT(Hr, N, j, p);
                EKF_AT(T2, N, i, j) = t;
            }
            __syncthreads();
            for (int e = tid; e < N * N; e += NTHREADS) {                  // S = Hr T2 + Rn (:324)
                const int i = e % N, j = e / N;
                double s = 0.0;
                for (int p = i; p < N; ++p) s += EKF_AT(Hr, N, i, p) * EKF_AT(T2, N, p, j);
                EKF_AT(S, N, i, j) = s + EKF_AT(Rn, N, i, j);
            }
            __syncthreads();
            ekf_cholesky<NTHREADS>(S, N, tid, &sh[3]);
            if (sh[3] >= 0) {
                status |= SLK_ST_SINGULAR;
            } else {
                // K = T2 S^-1 (:325): row i of K solves S x = T2(i, :)^T; one row per thread
                for (int i = tid; i < N; i += NTHREADS) {
                    for (int c = 0; c < N; ++c) {                          // forward: Ls y = t
                        double s = EKF_AT(T2, N, i, c);
                        for (int p = 0; p < c; ++p) s -= EKF_AT(S, N, c, p) * EKF_AT(K, N, i, p);
                        EKF_AT(K, N, i, c) = s / EKF_AT(S, N, c, c);
                    }
                    for (int c = N - 1; c >= 0; --c) {                     // backward: Ls^T x = y
                        double s = EKF_AT(K, N, i, c);
                        for (int p = c + 1; p < N; ++p) s -= EKF_AT(S, N, p, c) * EKF_AT(K, N, i, p);
                        EKF_AT(K, N, i, c) = s / EKF_AT(S, N, c, c);
                    }
                }
                __syncthreads();
                // Pk -= K S K^T = K T2^T (:330), delta = K innovation
                for (int e = tid; e < N * N; e += NTHREADS) {
                    const int i = e % N, j = e / N;
                    double s = 0.0;
                    for (int p = 0; p < N; ++p) s += EKF_AT(K, N, i, p) * EKF_AT(T2, N, j, p);
                    EKF_AT(Pn, N, i, j) = EKF_AT(P, N, i, j) - s;
                }
                for (int i = tid; i < N; i += NTHREADS) {
                    double s = 0.0;
                    for (int p = 0; p < N; ++p) s += EKF_AT(K, N, i, p) * rn[p];
                    delta[i] = s;
                }
                __syncthreads();
                for (int e = tid; e < N * N; e += NTHREADS) P[e] = Pn[e];
                // mu <- mu [+] delta (:331; MultiState boxplus, State.hpp:418-434)
                for (int blk = tid; blk <= a.k; blk += NTHREADS) {
                    const int to = blk ? 12 + 6 * (blk - 1) : 0, so = blk ? 13 + 7 * (blk - 1) : 0;
                    for (int c = 0; c < 3; ++c) mean[so + c] += delta[to + c];
                    stq(mean + so + 3, qmul(ldq(mean + so + 3), so3_exp(delta[to + 3], delta[to + 4], delta[to + 5])));
                    if (blk == 0) for (int c = 0; c < 6; ++c) mean[7 + c] += delta[6 + c];
                }
            }
        }
    }
    if (tid == 0 && status) atomicOr(a.status + b, status);
}

} // namespace slk
