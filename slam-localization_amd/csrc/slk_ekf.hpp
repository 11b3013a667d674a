// slk_ekf.hpp -- Msckf EKF update on the GPU (reference src/filters/Msckf.hpp:284-349).
//
// zmean = h(mu) and the Jacobian H (m x N) come from the caller's functor (:310).  One workgroup per filter; the
// dense work runs on a per-filter global workspace (it stays in L2 / Infinity Cache): this first version is the
// straightforward fp64 restatement of the reference's steps -- correctness and parity first, the GEMM-shaped parts
// (H P H^T, thinQ^T R thinQ, H P H^T + R, K S K^T) are the candidates for the matrix cores next.
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

// C(i, j) = sum_p A(i, p) B(p, j) on the fp64 matrix cores: 16x16 output tiles dealt to the waves, operands fetched
// per lane through the element functors (A fragment: row l&15, k l>>4; B fragment: k l>>4, column l&15), two k-steps
// per trip on independent accumulators.  store(row, col, value) sees every in-range element of the computed tiles.
template <class AFn, class BFn, class StoreFn>
__device__ __forceinline__ void ekf_mfma_gemm(int M, int Nc, int K, bool lower_only, int wave, int nwaves, int lane,
                                              AFn Ael, BFn Bel, StoreFn store)
{
    const int c = lane & 15, g = lane >> 4;
    const int ntr = (M + 15) >> 4, ntc = (Nc + 15) >> 4;
    for (int t = wave; t < ntr * ntc; t += nwaves) {
        const int I = t % ntr, J = t / ntr;
        if (lower_only && I < J) continue;
        const int r = 16 * I + c, cc = 16 * J + c;
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
        for (int k0 = 0; k0 < K; k0 += 8) {
            const int p0 = k0 + g, p1 = k0 + 4 + g;
            const double a0 = (r < M && p0 < K) ? Ael(r, p0) : 0.0, b0 = (cc < Nc && p0 < K) ? Bel(p0, cc) : 0.0;
            const double a1 = (r < M && p1 < K) ? Ael(r, p1) : 0.0, b1 = (cc < Nc && p1 < K) ? Bel(p1, cc) : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
        }
        acc0 = acc0 + acc1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 16 * I + g + 4 * q;
            if (row < M && cc < Nc) store(row, cc, acc0[q]);
        }
    }
}

// ------------------------------------------------------------------ LDS-resident variant (m <= 128, N <= 64)
// Same steps; the two factorisations, the inverse factor, the QR sweep and thinQ live in LDS (132 KB: one workgroup
// per CU), dots over a column are split over four threads.  Everything GEMM-shaped still reads the global workspace.
template <int NTHREADS>
__device__ void ekf_cholesky_packed_lds(double *A /* packed lower, column-major */, int n, int tid, int *flag)
{
    for (int j = 0; j < n; ++j) {
        const double d = A[pk(n, j, j)];
        if (!(d > 0.0)) { if (tid == 0 && *flag < 0) *flag = j; }
        __syncthreads();
        const double rs = 1.0 / sqrt(d);
        const int base = pk(n, j, j) - j;                       // A(i, j) = A[base + i]
        for (int i = j + 1 + tid; i < n; i += NTHREADS) A[base + i] *= rs;
        if (tid == 0) A[base + j] = sqrt(d);
        __syncthreads();
        // trailing lower triangle in 16 x 16 thread tiles
        constexpr int TJ = NTHREADS / 16;                       // thread tile: 16 rows x TJ columns
        const int ti = tid & 15, tj = tid >> 4, rem = n - j - 1, nbr = (rem + 15) >> 4, nbc = (rem + TJ - 1) / TJ;
        for (int bc = 0; bc < nbc; ++bc)
            for (int br = (TJ * bc) >> 4; br < nbr; ++br) {
                const int c = j + 1 + TJ * bc + tj, i = j + 1 + 16 * br + ti;
                if (i < n && c < n && i >= c) A[pk(n, i, c)] -= A[base + i] * A[base + c];
            }
        __syncthreads();
    }
}

// Blocked right-looking Cholesky on any lower-triangle accessor A(i, j) -> double& (packed or full storage, LDS):
// panels of 16 columns are factored column by column (the rank-1 updates touch the panel's own columns only), the
// trailing matrix gets one rank-16 update per panel on the matrix cores.
template <int NTHREADS, class AccFn>
__device__ __forceinline__ void ekf_cholesky_blocked(int n, int tid, int *flag, AccFn A)
{
    constexpr int NWV = NTHREADS / 64;
    const int wv = tid >> 6, ln = tid & 63;
    for (int J = 0; J < n; J += 16) {
        const int nb = (n - J < 16) ? n - J : 16;
        // (a) the 16 x 16 diagonal block, column by column inside ONE wave (wave-local LDS ordering, no barrier)
        if (wv == 0) {
            for (int jj = 0; jj < nb; ++jj) {
                const int j = J + jj;
                const double d = A(j, j);
                if (!(d > 0.0)) { if (ln == 0 && *flag < 0) *flag = j; }
                const double rs = 1.0 / sqrt(d);
                wave_sync();
                if (ln > jj && ln < nb) A(J + ln, j) *= rs;
                if (ln == jj) A(j, j) = sqrt(d);
                wave_sync();
                const int pc = nb - jj - 1;                          // remaining columns of the block
                for (int e = ln; e < pc * pc; e += 64) {
                    const int c = jj + 1 + e / pc, i = jj + 1 + e % pc;
                    if (i >= c) A(J + i, J + c) -= A(J + i, j) * A(J + c, j);
                }
                wave_sync();
            }
        }
        __syncthreads();
        // (b) rows below the block: X L11^T = A21, one row per thread, the row's 16 values in registers
        const int T0 = J + nb, rem2 = n - T0;
        for (int r = tid; r < rem2; r += NTHREADS) {
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c < nb) {
                    double v = A(T0 + r, J + c);
#pragma unroll
                    for (int p = 0; p < c; ++p) v -= x[p] * A(J + c, J + p);
                    x[c] = v / A(J + c, J + c);
                    A(T0 + r, J + c) = x[c];
                } else {
                    x[c] = 0.0;
                }
            }
        }
        __syncthreads();
        // (c) trailing matrix: one rank-16 update on the matrix cores
        if (rem2 > 0) {
            ekf_mfma_gemm(rem2, rem2, nb, true, wv, NWV, ln, [&](int i, int p) { return A(T0 + i, J + p); },
                          [&](int p, int c) { return A(T0 + c, J + p); },
                          [&](int i, int c, double v) { if (i >= c) A(T0 + i, T0 + c) -= v; });
            __syncthreads();
        }
    }
}

// Inverse of a lower-triangular factor, row by row: row i of L^-1 is -(1 / L_ii) * L(i, 0..i-1) * L^-1(0..i-1, :) --
// every column is an independent dot product, G lanes each, one workgroup barrier per row.
template <int NTHREADS, class LFn, class RdFn, class WrFn>
__device__ __forceinline__ void ekf_inverse_lower(int n, int tid, LFn Lel, RdFn rd, WrFn wr)
{
    constexpr int G = 8, CPP = NTHREADS / G;
    const int sub = tid % G, cl = tid / G;
    for (int i = 0; i < n; ++i) {
        const double rinv = 1.0 / Lel(i, i);
        for (int c0 = 0; c0 <= i; c0 += CPP) {
            const int c = c0 + cl;
            double sacc = 0.0;
            if (c < i) for (int p = c + sub; p < i; p += G) sacc += Lel(i, p) * rd(p, c);
            sacc += __shfl_xor(sacc, 1, 64);
            sacc += __shfl_xor(sacc, 2, 64);
            sacc += __shfl_xor(sacc, 4, 64);
            if (sub == 0 && c <= i) wr(i, c, (c == i) ? rinv : -sacc * rinv);
        }
        __syncthreads();
    }
}

template <int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void msckf_ekf_lds_kernel(EkfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NWV = NTHREADS / 64;
    const int wvi = threadIdx.x >> 6, lni = threadIdx.x & 63;
    __shared__ int idx[136];
    __shared__ int sh[8];
    __shared__ double hh[4];
    __shared__ double infob[3 * 64];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int N = a.N, Nq = a.Nq, m = a.m;
    double *mean = a.mean + (size_t)b * Nq, *P = a.P + (size_t)b * N * N;
    const double *z = a.z + (size_t)b * m, *zm = a.zmean + (size_t)b * m, *H = a.H + (size_t)b * m * N;
    const double *R = a.R + (size_t)b * a.r_stride;
    double *w = a.ws + (size_t)b * ekf_ws_doubles(N, m);
    double *PHt = w;                w += (size_t)N * m;
    w += 2 * (size_t)m * m;                                   // (S0, Li of the global variant: unused here)
    double *Rr = w;                 w += (size_t)m * m;
    w += 2 * (size_t)m * N;                                   // (Hq, Q1 of the global variant)
    double *T1 = w;                 w += (size_t)m * N;
    double *HrG = w;                w += (size_t)N * N;
    double *Rn = w;                 w += (size_t)N * N;
    w += 3 * (size_t)N * N;
    double *Pn = w;                 w += (size_t)N * N;
    double *innov = w;              w += m;
    double *rq = w;                 w += m;
    double *tau = w;                w += 2 * m;
    double *rn = w;                 w += N;
    double *delta = w;              w += 3 * N;
    const int np = pk_size(m);
    double *L0 = lds, *Li = lds + np;                          // gate phase: packed factor and its inverse
    if (tid == 0) { sh[0] = m; sh[1] = 0; sh[2] = -1; sh[3] = -1; a.outliers[b] = 0u; }
    EKF_STAMP(0);
    for (int r = tid; r < m; r += NTHREADS) { innov[r] = z[r] - zm[r]; idx[r] = r; }
    ekf_mfma_gemm(N, m, N, false, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(P, N, i, p); },
                  [&](int p, int j) { return EKF_AT(H, m, j, p); }, [&](int i, int j, double v) { EKF_AT(PHt, N, i, j) = v; });
    __syncthreads();
    ekf_mfma_gemm(m, m, N, true, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(H, m, i, p); },     // lower triangle of S0
                  [&](int p, int j) { return EKF_AT(PHt, N, p, j); },
                  [&](int i, int j, double v) { if (i >= j) L0[pk(m, i, j)] = v + EKF_AT(R, m, i, j); });
    __syncthreads();
    EKF_STAMP(1);
    ekf_cholesky_blocked<NTHREADS>(m, tid, &sh[2], [&](int i, int j) -> double & { return L0[pk(m, i, j)]; });
    EKF_STAMP(2);
    int status = 0;
    if (sh[2] >= 0) {
        status |= SLK_ST_SINGULAR;
    } else {
        ekf_inverse_lower<NTHREADS>(m, tid, [&](int i, int p) { return L0[pk(m, i, p)]; },       // Li = L0^-1
                                    [&](int p, int c) { return Li[pk(m, p, c)]; },
                                    [&](int i, int c, double v) { Li[pk(m, i, c)] = v; });
        EKF_STAMP(3);
        for (int e = tid; e < 3 * (m / 2); e += NTHREADS) {    // the 2x2 diagonal blocks of the information matrix
            const int blk = e / 3, q = e % 3, r = 2 * blk + (q == 2), c = 2 * blk + (q >= 1);
            double s = 0.0;
            for (int p = (r > c ? r : c); p < m; ++p) s += Li[pk(m, p, r)] * Li[pk(m, p, c)];
            infob[e] = s;                                      // q = 0: (0,0), 1: (0,1), 2: (1,1)
        }
        __syncthreads();
        if (tid == 0) {
            int cnt = m;
            unsigned nout = 0;
            int i = 0;
            while (i < cnt / 2) {
                const double r0 = innov[idx[2 * i]], r1 = innov[idx[2 * i + 1]];
                const double i00 = infob[3 * i], i01 = infob[3 * i + 1], i11 = infob[3 * i + 2];
                const double d2 = r0 * (i00 * r0 + i01 * r1) + r1 * (i01 * r0 + i11 * r1);
                const bool ok = a.gate ? (d2 < 5.99) : true;
                if (!ok) {
                    for (int rep = 0; rep < 2; ++rep) {
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
            a.outliers[b] = nout;
        }
        __syncthreads();
        const int mm = sh[0];
        EKF_STAMP(4);
        if (mm > 0 && mm < N) {
            status |= SLK_ST_EKF_ROWS;
        } else if (mm > 0) {
            double *Hq = lds, *Q1 = lds + (size_t)mm * N;      // QR phase: both mm x N, column-major
            for (int e = tid; e < mm * N; e += NTHREADS) { const int i = e % mm, j = e / mm; EKF_AT(Hq, mm, i, j) = EKF_AT(H, m, idx[i], j); }
            for (int e = tid; e < mm * mm; e += NTHREADS) { const int i = e % mm, j = e / mm; EKF_AT(Rr, mm, i, j) = EKF_AT(R, m, idx[i], idx[j]); }
            for (int i = tid; i < mm; i += NTHREADS) rq[i] = innov[idx[i]];
            __syncthreads();
            EKF_STAMP(5);
            constexpr int G = 16, CPP = NTHREADS / G;          // sixteen lanes per column
            const int colg = tid / G, part = tid % G;
            for (int kk = 0; kk < N; ++kk) {
                if (tid < 64) {                                // tail norm of column kk by one wave
                    double part2 = 0.0;
                    for (int i = kk + 1 + tid; i < mm; i += 64) { const double v = EKF_AT(Hq, mm, i, kk); part2 += v * v; }
                    for (int o = 32; o >= 1; o >>= 1) part2 += __shfl_xor(part2, o, 64);
                    if (tid == 0) {
                        const double c0 = EKF_AT(Hq, mm, kk, kk);
                        double beta, tk, den;
                        if (part2 <= 2.2250738585072014e-308) { tk = 0.0; beta = c0; den = 0.0; }
                        else {
                            beta = sqrt(c0 * c0 + part2);
                            if (c0 >= 0.0) beta = -beta;
                            den = c0 - beta;
                            tk = (beta - c0) / beta;
                        }
                        hh[0] = beta; hh[1] = tk; hh[2] = den;
                        tau[kk] = tk;
                    }
                }
                __syncthreads();
                const double tk = hh[1], den = hh[2];
                for (int i = kk + 1 + tid; i < mm; i += NTHREADS) EKF_AT(Hq, mm, i, kk) = (den != 0.0) ? EKF_AT(Hq, mm, i, kk) / den : 0.0;
                if (tid == 0) EKF_AT(Hq, mm, kk, kk) = hh[0];
                __syncthreads();
                for (int j0 = kk + 1; j0 < N; j0 += CPP) {
                    const int j = j0 + colg;
                    double wv = 0.0;
                    if (j < N) {
                        if (part == 0) wv = EKF_AT(Hq, mm, kk, j);
                        for (int i = kk + 1 + part; i < mm; i += G) wv += EKF_AT(Hq, mm, i, kk) * EKF_AT(Hq, mm, i, j);
                    }
                    wv += __shfl_xor(wv, 1, 64);
                    wv += __shfl_xor(wv, 2, 64);
                    wv += __shfl_xor(wv, 4, 64);
                    wv += __shfl_xor(wv, 8, 64);
                    wv *= tk;
                    if (j < N) {
                        if (part == 0) EKF_AT(Hq, mm, kk, j) -= wv;
                        for (int i = kk + 1 + part; i < mm; i += G) EKF_AT(Hq, mm, i, j) -= EKF_AT(Hq, mm, i, kk) * wv;
                    }
                }
                __syncthreads();
            }
            EKF_STAMP(6);
            // thinQ: sixteen lanes (of one wave) per column, reflectors in reverse; a column is private to its lanes
            for (int j0 = 0; j0 < N; j0 += CPP) {
                const int j = j0 + colg;
                if (j < N) for (int i = part; i < mm; i += G) EKF_AT(Q1, mm, i, j) = (i == j) ? 1.0 : 0.0;
                for (int kk = N - 1; kk >= 0; --kk) {
                    double wv = 0.0;
                    if (j < N) {
                        if (part == 0) wv = EKF_AT(Q1, mm, kk, j);
                        for (int i = kk + 1 + part; i < mm; i += G) wv += EKF_AT(Hq, mm, i, kk) * EKF_AT(Q1, mm, i, j);
                    }
                    wv += __shfl_xor(wv, 1, 64);
                    wv += __shfl_xor(wv, 2, 64);
                    wv += __shfl_xor(wv, 4, 64);
                    wv += __shfl_xor(wv, 8, 64);
                    wv *= tau[kk];
                    if (j < N) {
                        if (part == 0) EKF_AT(Q1, mm, kk, j) -= wv;
                        for (int i = kk + 1 + part; i < mm; i += G) EKF_AT(Q1, mm, i, j) -= EKF_AT(Hq, mm, i, kk) * wv;
                    }
                }
            }
            __syncthreads();
            EKF_STAMP(7);
            for (int e = tid; e < N * N; e += NTHREADS) { const int i = e % N, j = e / N; EKF_AT(HrG, N, i, j) = (i <= j) ? EKF_AT(Hq, mm, i, j) : 0.0; }
            for (int j = tid; j < N; j += NTHREADS) {
                double s = 0.0;
                for (int i = 0; i < mm; ++i) s += EKF_AT(Q1, mm, i, j) * rq[i];
                rn[j] = s;
            }
            ekf_mfma_gemm(mm, N, mm, false, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(Rr, mm, i, p); },   // T1 = Rr thinQ
                          [&](int p, int j) { return EKF_AT(Q1, mm, p, j); }, [&](int i, int j, double v) { EKF_AT(T1, mm, i, j) = v; });
            __syncthreads();
            ekf_mfma_gemm(N, N, mm, false, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(Q1, mm, p, i); },    // Rn = thinQ^T T1
                          [&](int p, int j) { return EKF_AT(T1, mm, p, j); }, [&](int i, int j, double v) { EKF_AT(Rn, N, i, j) = v; });
            __syncthreads();
            EKF_STAMP(8);
            // gain phase: Hr, T2, S, K in LDS (N x N each)
            double *Hr = lds, *T2 = Hr + (size_t)N * N, *S = T2 + (size_t)N * N, *K = S + (size_t)N * N;
            for (int e = tid; e < N * N; e += NTHREADS) Hr[e] = HrG[e];
            __syncthreads();
            ekf_mfma_gemm(N, N, N, false, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(P, N, i, p); },       // T2 = P Hr^T
                          [&](int p, int j) { return EKF_AT(Hr, N, j, p); }, [&](int i, int j, double v) { EKF_AT(T2, N, i, j) = v; });
            __syncthreads();
            ekf_mfma_gemm(N, N, N, false, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(Hr, N, i, p); },      // S = Hr T2 + Rn
                          [&](int p, int j) { return EKF_AT(T2, N, p, j); },
                          [&](int i, int j, double v) { EKF_AT(S, N, i, j) = v + EKF_AT(Rn, N, i, j); });
            __syncthreads();
            EKF_STAMP(9);
            ekf_cholesky_blocked<NTHREADS>(N, tid, &sh[3], [&](int i, int j) -> double & { return EKF_AT(S, N, i, j); });
            EKF_STAMP(10);
            if (sh[3] >= 0) {
                status |= SLK_ST_SINGULAR;
            } else {
                // K = T2 S^-1 = (T2 Ls^-T) Ls^-1: the inverse factor row by row, then two products on the matrix cores.
                // Slots: Hr (dead) takes Ls^-1, K takes Y = T2 Ls^-T, S (dead once inverted) takes the final K.
                double *Lsi = Hr, *Y = K, *Kf = S;
                ekf_inverse_lower<NTHREADS>(N, tid, [&](int i, int p) { return EKF_AT(S, N, i, p); },
                                            [&](int p, int c) { return EKF_AT(Lsi, N, p, c); },
                                            [&](int i, int c, double v) { EKF_AT(Lsi, N, i, c) = v; });
                ekf_mfma_gemm(N, N, N, false, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(T2, N, i, p); },       // Y = T2 Ls^-T
                              [&](int p, int j) { return (p <= j) ? EKF_AT(Lsi, N, j, p) : 0.0; },
                              [&](int i, int j, double v) { EKF_AT(Y, N, i, j) = v; });
                __syncthreads();
                ekf_mfma_gemm(N, N, N, false, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(Y, N, i, p); },        // K = Y Ls^-1
                              [&](int p, int j) { return (p >= j) ? EKF_AT(Lsi, N, p, j) : 0.0; },
                              [&](int i, int j, double v) { EKF_AT(Kf, N, i, j) = v; });
                __syncthreads();
                EKF_STAMP(11);
                ekf_mfma_gemm(N, N, N, false, wvi, NWV, lni, [&](int i, int p) { return EKF_AT(Kf, N, i, p); },  // Pk - K T2^T
                              [&](int p, int j) { return EKF_AT(T2, N, j, p); },
                              [&](int i, int j, double v) { EKF_AT(Pn, N, i, j) = EKF_AT(P, N, i, j) - v; });
                for (int i = tid; i < N; i += NTHREADS) {
                    double s = 0.0;
                    for (int p = 0; p < N; ++p) s += EKF_AT(Kf, N, i, p) * rn[p];
                    delta[i] = s;
                }
                __syncthreads();
                for (int e = tid; e < N * N; e += NTHREADS) P[e] = Pn[e];
                for (int blk = tid; blk <= a.k; blk += NTHREADS) {
                    const int to = blk ? 12 + 6 * (blk - 1) : 0, so = blk ? 13 + 7 * (blk - 1) : 0;
                    for (int c = 0; c < 3; ++c) mean[so + c] += delta[to + c];
                    stq(mean + so + 3, qmul(ldq(mean + so + 3), so3_exp(delta[to + 3], delta[to + 4], delta[to + 5])));
                    if (blk == 0) for (int c = 0; c < 6; ++c) mean[7 + c] += delta[6 + c];
                }
            }
        }
    }
    EKF_STAMP(12);
    if (tid == 0 && status) atomicOr(a.status + b, status);
}

// LDS doubles the resident variant needs
__host__ __device__ inline size_t ekf_lds_doubles(int N, int m)
{
    size_t g = 2 * (size_t)pk_size(m), q = 2 * (size_t)m * N, k = 4 * (size_t)N * N;
    size_t r = g > q ? g : q;
    return r > k ? r : k;
}

} // namespace slk
