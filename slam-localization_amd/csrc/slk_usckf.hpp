// slk_usckf.hpp -- HIP kernels for localization::Usckf (reference src/filters/Usckf.hpp):
// predict with clone / feature cross-covariance propagation (:107-244), UKF update with direct
// boxplus correction (:246-308), cloning (:391-433) and setMeasurement (:322-389).
// One workgroup per filter, covariance resident in LDS.
#pragma once
#include "slk_kernels.hpp"

namespace slk {

struct UCarve { int P, Lm, mu, small, colbuf, pool, total; int lda, S; };
__device__ __forceinline__ bool usckf_update_fast(const KArgs &a, double *smem);     // slk_usckf_fast.hpp

__host__ __device__ inline UCarve carve_usckf(int N, int Nq, int m, int NT, bool split = false)
{
    UCarve c;
    c.lda = split ? N : (N | 1);
    c.S = 2 * N + 1;
    int o = 0;
    c.P = o;      o += split ? 0 : round_up(N * c.lda, 2);     // (split path: the covariance stays in global memory)
    c.Lm = o;     o += round_up(pk_size(N), 2);
    c.mu = o;     o += round_up(Nq, 2);
    c.small = o;  o += 64;
    c.colbuf = o; o += 4 * ((16 * NT > 32 ? 16 * NT : 32) + 2);
    c.pool = o;
    int upd = round_up(c.S * m, 2) + 3 * round_up(N * m, 2) + round_up(m * m, 2) + round_up(m * (m + 1), 2)
              + 4 * round_up(m, 2) + round_up(N, 2);
    int pred = split ? 0 : PRED_SCRATCH + 160 + 160 + 2 * round_up(12 * N, 2);
    c.total = o + (upd > pred ? upd : pred);
    return c;
}

// SPLIT (N <= 64, three launches per step like the Msckf path): predict runs in usckf_predict_kernel, the factorisation in
// msckf_chol_kernel (one wave per filter each), and this kernel is the update alone: the covariance stays in global
// memory (its diagonal for the moments, the downdate as a read-modify-write), the factor comes from the workspace.
// UEX: exact-shape instantiation of the unit-test layout (UsckfUnitTest.cpp: 3 + 9 features, N = 48, m = 3 rows): layout
// and packed-index arithmetic fold.
template <int NT, int NTHREADS, bool SPLIT = false, bool UEX = false>
__global__ __launch_bounds__(NTHREADS) void usckf_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NW = NTHREADS / 64;
    constexpr int GD = Grid<NTHREADS>::GD;
    constexpr int SDN = (16 * NT + GD - 1) / GD;
    constexpr int SDM = (MAXM + GD - 1) / GD;
    const int bidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Lay L = a.lay;
    if constexpr (UEX) { L.kind = SLK_USCKF; L.nfk = 3; L.nfkl = 9; L.N = 48; L.Nq = 51; L.nso3 = 3; a.m = 3; }
    const int N = L.N, Nq = L.Nq, m = a.m;
    const UCarve cv = carve_usckf(N, Nq, m, NT, SPLIT);
    const int lda = cv.lda, S = cv.S;
    double *gmean = a.mean + (size_t)bidx * Nq;
    double *gP = a.P + (size_t)bidx * N * N;
    double *P = SPLIT ? gP : smem + cv.P, *Lm = smem + cv.Lm, *mu = smem + cv.mu, *colbuf = smem + cv.colbuf, *pool = smem + cv.pool;
    int *ish = reinterpret_cast<int *>(smem + cv.small);
    int status = 0;
#ifndef SLK_NO_FAST_STEP
    if constexpr (SPLIT && UEX && NTHREADS == 128) {
        if (usckf_update_fast(a, smem)) return;                 // (slk_usckf_fast.hpp; false: before any global write)
    }
#endif
    if (a.do_update && a.emit != 4 && tid == 0) a.outliers[bidx] = 0u;
    if (tid == 0) ish[42] = 0;

    SLK_STAMP_NR(0);
    for (int e = tid; e < Nq; e += NTHREADS) mu[e] = gmean[e];
    if constexpr (!SPLIT)
        for (int c = wave; c < N; c += NW)
            for (int r = lane; r < N; r += 64) P[r + c * lda] = gP[r + (size_t)c * N];
    __syncthreads();

    if (!SPLIT && (a.do_predict || a.emit == 1)) {
        // ---- Usckf::predict, Usckf.hpp:107-244
        double *Lblk = pool, *Pn = pool + 160, *Pxy = pool + 320, *Fk = pool + 480, *scr = pool + 640;
        double *RB = pool + 640 + 800;             // old rows 24..35 of P: 12 x N (ld 12)
        double *CB = RB + round_up(12 * N, 2);     // old cols 24..35 of P: N x 12 (ld N)
        if (wave == 0) {
            int st0 = predict_phase<true>(a, bidx, tid, [&](int i, int j) { return P[(24 + i) + (24 + j) * lda]; },
                                          Lblk, mu + 26, Pn, scr, Pxy);
            if (tid == 0) ish[44] = st0;
        }
        __syncthreads();
        const int st = ish[44];
        if (a.emit == 1) return;
        status |= st;
        if (!(st & SLK_ST_LLT_FAIL)) {
            // Fk = Pxy^T * Pk_i^-1 (:154).  Pk_i = L L^T (its Cholesky factor is in Lblk) and Pxy = L M (predict_phase hands
            // out M): Fk^T = L^-T M, one backward substitution per column.
            if (tid < 12) {
                double x[12];
                for (int r = 11; r >= 0; --r) {
                    double s = Pxy[r + 12 * tid];
                    for (int p = r + 1; p < 12; ++p) s -= Lblk[pk(12, p, r)] * x[p];
                    x[r] = s / Lblk[pk(12, r, r)];
                }
                for (int r = 0; r < 12; ++r) Fk[tid + 12 * r] = x[r];      // column tid of Fk^T = row tid of Fk
            }
            for (int e = tid; e < 12 * N; e += NTHREADS) {
                int r = e % 12, c = e / 12;
                RB[e] = P[(24 + r) + c * lda];
                int rr = e % N, cc = e / N;
                CB[e] = P[rr + (24 + cc) * lda];
            }
            __syncthreads();
            // rows of state k+i against statek, statek_l and both feature blocks: Fk * block (:200-208, :221-232)
            for (int e = tid; e < 12 * N; e += NTHREADS) {
                int r = e % 12, c = e / 12;
                if (c >= 24 && c < 36) continue;
                double s = 0.0;
                for (int p = 0; p < 12; ++p) s += Fk[r + 12 * p] * RB[p + 12 * c];
                P[(24 + r) + c * lda] = s;
            }
            // columns of state k+i against statek and statek_l: block * Fk^T (:190-198)
            for (int e = tid; e < 24 * 12; e += NTHREADS) {
                int r = e % 24, c = e / 24;
                double s = 0.0;
                for (int p = 0; p < 12; ++p) s += CB[r + N * p] * Fk[c + 12 * p];
                P[r + (24 + c) * lda] = s;
            }
            for (int e = tid; e < 144; e += NTHREADS) { int r = e % 12, c = e / 12; P[(24 + r) + (24 + c) * lda] = Pn[e]; }
            __syncthreads();
            // feature rows against state k+i = transposes of the updated blocks (:227, :235)
            for (int e = tid; e < (N - 36) * 12; e += NTHREADS) {
                int j = 36 + e % (N - 36), c = e / (N - 36);
                P[j + (24 + c) * lda] = P[(24 + c) + j * lda];
            }
            __syncthreads();
            if (!a.do_update) {
                for (int c = wave; c < N; c += NW)
                    for (int r = lane; r < N; r += 64) gP[r + (size_t)c * N] = P[r + c * lda];
                for (int e = tid; e < Nq; e += NTHREADS) gmean[e] = mu[e];
            }
        }
        __syncthreads();
    }

    if (a.do_update || a.emit == 2) {
        // ---- Usckf::update, Usckf.hpp:246-308
        int fail;
        if (SPLIT && NT <= 4 && !a.wsfail) {
            // no factor in the workspace (the exact-shape launch factors inside the update kernel, slk_usckf_fast.hpp, and
            // this body is its fallback): one wave, panel by rows, straight into LDS
            if constexpr (NT <= 4) {
                if (wave == 0) {
                    d4 acc[CholM<NT>::NTL];
                    cholm_load_t<NT>(acc, N, lane, [&](int i, int j) { return gP[i + (size_t)j * N]; });
                    const int f0 = cholp_factor<NT, 0>(acc, Lm, N, colbuf, lane);
                    if (lane == 0) ish[45] = f0;
                }
            }
            __syncthreads();
            fail = ish[45];
        } else if constexpr (SPLIT) {
            const double *gL = a.wsL + (size_t)bidx * pk_size(N);
            for (int e0 = 0; e0 < pk_size(N); e0 += 12 * NTHREADS) {      // twelve loads in flight per thread
                double v[12];
#pragma unroll
                for (int q = 0; q < 12; ++q) { const int e = e0 + q * NTHREADS + tid; v[q] = (e < pk_size(N)) ? gL[e] : 0.0; }
#pragma unroll
                for (int q = 0; q < 12; ++q) { const int e = e0 + q * NTHREADS + tid; if (e < pk_size(N)) Lm[e] = v[q]; }
            }
            fail = a.wsfail[bidx];
            __syncthreads();
        } else if constexpr (NT <= 4) {
            if (wave == 0) {
                d4 acc[CholM<NT>::NTL];
                cholm_load<NT>(acc, N, lane, [&](int i, int j) { return P[i + j * lda]; });
                int f0 = cholm_factor<NT>(acc, Lm, N, colbuf, lane);
                if (lane == 0) ish[45] = f0;
            }
            __syncthreads();
            fail = ish[45];
        } else {
            fail = chol_packed<NTHREADS, SDN>(Lm, N, colbuf, tid, [&](int i, int j) { return P[i + j * lda]; });
        }
        bool applied = false;
        SLK_STAMP_NR(3);
        if (fail >= 0) {
            status |= SLK_ST_LLT_FAIL;
        } else if (a.emit == 2) {
            double *X = a.Xout + (size_t)bidx * S * Nq;
            for (int e = tid; e < S * N; e += NTHREADS) {
                int t = e % N, i = e / N, blk = 0, comp = 0;
                int s = t2s(L, t, blk, comp);
                if (s >= 0) X[(size_t)i * Nq + s] = mu[s] + pert(Lm, N, nullptr, t, sig_of(i));
            }
            for (int e = tid; e < S * 3; e += NTHREADS) {
                int b = e % 3, i = e / 3;
                Quat q = sigma_quat(L, mu, Lm, nullptr, b, sig_of(i));
                double *o = X + (size_t)i * Nq + so3_soff(L, b);
                o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
            }
        } else if (!pose_params_ok(a, L, a.mp ? a.mp + (size_t)bidx * a.mp_stride : nullptr)) {
            status |= SLK_ST_BAD_INDEX;                       // pose index out of 0..2: update skipped
        } else {
            double *Z = pool;
            double *DZ = Z + round_up(S * m, 2);
            double *Pxz = DZ + round_up(N * m, 2);
            double *K = Pxz + round_up(N * m, 2);
            double *Sm = K + round_up(N * m, 2);
            double *G = Sm + round_up(m * m, 2);                 // Cholesky factor of S (ld m+1)
            double *zbar = G + round_up(m * (m + 1), 2);
            double *innov = zbar + round_up(m, 2);
            double *wv = innov + round_up(m, 2);                 // Ls^-1 innovation
            double *dlt = wv + 2 * round_up(m, 2);
            measurement_moments<NTHREADS>(a, L, bidx, tid, mu, Lm, Z, DZ, Pxz, Sm, zbar, innov, &ish[42],
                                           [&](int t) { return P[t + t * lda]; });
            SLK_STAMP_NR(6);
            if (a.emit == 4) {
                // innovation and its covariance for a caller-side significance test (Usckf.hpp:262-302 with an
                // arbitrary `mt`): Xout [B][m*m + m] = S (column-major), innovation; nothing else happens
                double *o = a.Xout + (size_t)bidx * (m * m + m);
                for (int e = tid; e < m * m; e += NTHREADS) o[e] = Sm[e];
                for (int e = tid; e < m; e += NTHREADS) o[m * m + e] = innov[e];
            }
            // S^-1 (:285-286): S = 1/2 dZ dZ^T + R is SPD for a valid R -> Cholesky, row-wise solves
            if (a.emit != 4 && wave == 0) {
                auto sel = [&](int i, int j) { return Sm[i + m * j]; };
                int f0;
                if (m <= 16) {
                    d4 acc[CholM<1>::NTL];
                    cholm_load<1>(acc, m, lane, sel);
                    f0 = cholm_factor<1>(acc, G, m, colbuf, lane);
                } else {
                    d4 acc[CholM<2>::NTL];
                    cholm_load<2>(acc, m, lane, sel);
                    f0 = cholm_factor<2>(acc, G, m, colbuf, lane);
                }
                if (lane == 0) ish[46] = f0;
            }
            __syncthreads();
            SLK_STAMP_NR(7);
            const int sfail = ish[46];
            if (a.emit == 4) {
            } else if (sfail >= 0) {
                status |= SLK_ST_SINGULAR;
            } else {
                for (int t = tid; t < N; t += NTHREADS) {             // K = covXZ * S^-1 :288
                    for (int c = 0; c < m; ++c) {
                        double sum = Pxz[t + N * c];
                        for (int p = 0; p < c; ++p) sum -= G[pk(m, c, p)] * K[t + N * p];
                        K[t + N * c] = sum / G[pk(m, c, c)];
                    }
                    for (int c = m - 1; c >= 0; --c) {
                        double sum = K[t + N * c];
                        for (int p = c + 1; p < m; ++p) sum -= G[pk(m, p, c)] * K[t + N * p];
                        K[t + N * c] = sum / G[pk(m, c, c)];
                    }
                }
                if (tid == 0) {                                        // mahalanobis2 = |Ls^-1 innovation|^2 :292
                    double d2 = 0.0;
                    for (int c = 0; c < m; ++c) {
                        double sum = innov[c];
                        for (int p = 0; p < c; ++p) sum -= G[pk(m, c, p)] * wv[p];
                        wv[c] = sum / G[pk(m, c, c)];
                        d2 += wv[c] * wv[c];
                    }
                    bool ok = true;
                    if (a.gate > 0) {
                        const double thr[10] = {0, 3.84, 5.99, 7.81, 9.49, 11.07, 12.59, 14.07, 15.51, 16.92};
                        ok = (a.gate <= 9) ? (d2 < thr[a.gate]) : false;   // Usckf.hpp:794-855
                    }
                    ish[40] = ok ? 1 : 0;
                }
                __syncthreads();
                SLK_STAMP_NR(8);
                if (!ish[40]) {
                    if (tid == 0) a.outliers[bidx] = 1u;
                    status |= SLK_ST_ALL_REJECTED;
                } else {
                    for (int t = tid; t < N; t += NTHREADS) {
                        double sum = 0.0;
                        for (int c = 0; c < m; ++c) sum += K[t + N * c] * innov[c];
                        dlt[t] = sum;
                    }
                    // Pk -= K S K^T (:296); K S = covXZ
                    if constexpr (SPLIT) {
                        // read-modify-write of the covariance in global memory: its loads first, all in flight
                        constexpr int DQ = (NTHREADS >= 256) ? 8 : 18;          // 48 x 48 at two waves: one batch
                        for (int e0 = 0; e0 < N * N; e0 += DQ * NTHREADS) {
                            double pv[DQ];
#pragma unroll
                            for (int q = 0; q < DQ; ++q) {
                                const int e = e0 + q * NTHREADS + tid;
                                pv[q] = (e < N * N) ? P[e] : 0.0;
                            }
#pragma unroll
                            for (int q = 0; q < DQ; ++q) {
                                const int e = e0 + q * NTHREADS + tid;
                                if (e < N * N) {
                                    const int i = e % N, j = e / N;
                                    double sum = 0.0;
                                    for (int c = 0; c < m; ++c) sum += Pxz[i + N * c] * K[j + N * c];
                                    P[e] = pv[q] - sum;
                                }
                            }
                        }
                    } else {
                        for (int j = wave; j < N; j += NW)
                            for (int i = lane; i < N; i += 64) {
                                double sum = 0.0;
                                for (int c = 0; c < m; ++c) sum += Pxz[i + N * c] * K[j + N * c];
                                P[i + j * lda] -= sum;
                            }
                    }
                    __syncthreads();
                    SLK_STAMP_NR(9);
                    // mu_state = mu_state + state(K * innovation) :299-301 (set() then boxplus through
                    // getVectorizedState(): exp/log round trip == direct boxplus for |rotation| < pi)
                    for (int t = tid; t < N; t += NTHREADS) {
                        int blk = 0, comp = 0, s = t2s(L, t, blk, comp);
                        if (s >= 0) mu[s] = mu[s] + dlt[t];
                    }
                    for (int b = tid; b < 3; b += NTHREADS) {
                        int to = so3_toff(L, b), so = so3_soff(L, b);
                        stq(mu + so, qmul(ldq(mu + so), so3_exp(dlt[to], dlt[to + 1], dlt[to + 2])));
                    }
                    applied = true;
                }
            }
        }
        __syncthreads();
        if (a.emit != 2 && a.emit != 4 && (applied || a.do_predict)) {
            if constexpr (!SPLIT)
                for (int c = wave; c < N; c += NW)
                    for (int r = lane; r < N; r += 64) gP[r + (size_t)c * N] = P[r + c * lda];
            if (applied || !SPLIT)
                for (int e = tid; e < Nq; e += NTHREADS) gmean[e] = mu[e];
        }
    }
    SLK_STAMP_NR(10);
    if (tid == 0 && status) atomicOr(a.status + bidx, status);
}

// Usckf::predict (Usckf.hpp:107-244) for the split path: one wave per filter, the covariance in global memory.  The
// 12-DOF prediction of state k+i (predict_phase), Fk = Pxy^T Pk_i^-1 (:154), then the cross blocks: rows of state k+i
// against everything else Fk * block (:200-208, :221-232), columns against statek / statek_l block * Fk^T (:190-198),
// feature rows as transposes (:227, :235).  All old values are staged in LDS before the first write.
#ifndef SLK_UPRED_WAVES
#define SLK_UPRED_WAVES 4
#endif
__global__ __launch_bounds__(64, SLK_UPRED_WAVES) void usckf_predict_kernel(KArgs a)
{
    // 9.9 KB and 128 registers: sixteen filters per CU, 4096 filters are ONE round (round 2: 13.4 KB, 190 registers, eight
    // filters per CU, two rounds of the same dependent chain).  Fk and the old rows RB live in the predict scratch (dead
    // once predict_phase returns), the old columns CB in the place of the 12 x 12 factor and of Pxy (dead once Fk stands).
    __shared__ __attribute__((aligned(16))) double sm[16 + 3 * 160 + 736];
    const int bidx = blockIdx.x, tid = threadIdx.x;
    const int N = a.lay.N, Nq = a.lay.Nq;
    double *mu = sm, *Pn = sm + 16, *Lblk = Pn + 160, *Pxy = Lblk + 160, *scr = Pxy + 160;
    double *RB = scr, *Fk = scr + 580;              // 12 x N (N <= 48: 576) + 144 <= 736
    double *gmean = a.mean + (size_t)bidx * Nq;
    double *gP = a.P + (size_t)bidx * N * N;
    if (tid < 13) mu[tid] = gmean[26 + tid];
    wave_sync();
    const int st = predict_phase<true>(a, bidx, tid, [&](int i, int j) { return gP[(24 + i) + (size_t)(24 + j) * N]; },
                                       Lblk, mu, Pn, scr, Pxy);
    if (st < 0) return;                              // sigma points emitted
    if (!(st & SLK_ST_LLT_FAIL)) {
        // the old rows 24..35 (12 x N, ld 12) to LDS before the first write (the predict scratch is dead), read from the LOWER
        // triangle only (columns beyond the block: the transposed entries, fetched along their columns); the old columns of
        // state k+i against statek / statek_l are the transposes of the first 24 of these rows -- no fetch of their own
        {
            double ra[7], rb[3];
#pragma unroll
            for (int q = 0; q < 7; ++q) {                                      // columns 0 .. 35: down the columns (inside the block: the lower triangle)
                const int e = tid + 64 * q, p = e % 12, c = e / 12, row = 24 + p;
                ra[q] = (e < 12 * 36) ? gP[c > row ? c + (size_t)row * N : row + (size_t)c * N] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {                                      // columns 36 .. N - 1: P(c, 24 + p), c fastest
                const int e = tid + 64 * q, cq = e % 12, p = e / 12;
                rb[q] = (e < 144 && 36 + cq < N) ? gP[(36 + cq) + (size_t)(24 + p) * N] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 7; ++q) { const int e = tid + 64 * q; if (e < 12 * 36) RB[e] = ra[q]; }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = tid + 64 * q, cq = e % 12, p = e / 12;
                if (e < 144 && 36 + cq < N) RB[p + 12 * (36 + cq)] = rb[q];
            }
        }
        double fk[12];                               // lanes < 12: column tid of Fk^T = row tid of Fk
        if (tid < 12) {                              // Fk^T = Pk_i^-1 Pxy = L^-T M (predict_phase hands out M): backward substitution per column
#pragma unroll
            for (int r = 11; r >= 0; --r) {
                double s = Pxy[r + 12 * tid];
#pragma unroll
                for (int p = r + 1; p < 12; ++p) s -= Lblk[pk(12, p, r)] * fk[p];
                const double dgl = Lblk[pk(12, r, r)];
                double rd = __builtin_amdgcn_rcp(dgl);       // reciprocal by two Newton steps (to about an ulp) instead of the division sequence
                rd = fma(fma(-dgl, rd, 1.0), rd, rd);
                rd = fma(fma(-dgl, rd, 1.0), rd, rd);
                fk[r] = s * rd;
                __builtin_amdgcn_sched_barrier(0);           // (one row of the factor in flight at a time: registers)
            }
        }
        wave_sync();
        if (tid < 12) {
#pragma unroll
            for (int r = 0; r < 12; ++r) Fk[tid + 12 * r] = fk[r];
        }
        wave_sync();
        {
            // The cross blocks on the matrix cores, transposed so that a lane's results run down a column of P: one
            // fragment of Fk (lane: Fk(c16, 4 ks + g)) serves both products.
            const int c16 = tid & 15, g = tid >> 4;
            double fa[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) { const double v = Fk[(c16 < 12 ? c16 : 0) + 12 * (4 * ks + g)]; fa[ks] = (c16 < 12) ? v : 0.0; }
            // rows of state k+i against everything but itself: (Fk * old rows)^T = RB^T Fk^T, tile T = columns 16 T .. of P
#pragma unroll
            for (int T = 0; T < 3; ++T) {
                if (16 * T < N) {
                    const int col = 16 * T + c16;
                    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        const double v = RB[(4 * ks + g) + 12 * (col < N ? col : 0)];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((col < N) ? v : 0.0, fa[ks], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cc = 16 * T + g + 4 * r;               // column of P; this lane's row is 24 + c16
                        if (c16 < 12 && cc < N && !(cc >= 24 && cc < 36)) {
                            if (cc < 24 || !a.lower_only) gP[(24 + c16) + (size_t)cc * N] = acc[r];
                            if (cc >= 36) gP[cc + (size_t)(24 + c16) * N] = acc[r];      // feature rows against state k+i: the transposes
                        }
                    }
                }
            }
            // columns of state k+i against statek and statek_l: (old cols * Fk^T)^T = Fk CB^T -- the transposes of the rows
            // above (nothing to do when only the lower triangle is kept up to date)
#pragma unroll
            for (int T = 0; T < 2 && !a.lower_only; ++T) {
                const int row = 16 * T + c16;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const double v = RB[(4 * ks + g) + 12 * (row < 24 ? row : 0)];      // old column entry (row, 24 + k) = old row entry (24 + k, row)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks], (row < 24) ? v : 0.0, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (row < 24) gP[row + (size_t)(24 + g + 4 * r) * N] = acc[r];
            }
        }
        for (int e = tid; e < 144; e += 64) gP[(24 + e % 12) + (size_t)(24 + e / 12) * N] = Pn[e];
        if (tid < 13) gmean[26 + tid] = mu[tid];
    }
    if (tid == 0 && st) atomicOr(a.status + bidx, st);
}

// Usckf::cloning, Usckf.hpp:391-433; one workgroup per filter.  Blocks: 0 statek, 1 statek_l, 2 statek_i
__global__ void usckf_cloning_kernel(double *mean, double *P, int B, int N, int Nq, int mode)
{
    int b = blockIdx.x, tid = threadIdx.x;
    if (b >= B) return;
    double *m = mean + (size_t)b * Nq, *p = P + (size_t)b * N * N;
    auto at = [&](int br, int bc, int i, int j) -> double & { return p[(12 * br + i) + (size_t)(12 * bc + j) * N]; };
    if (tid < 144) {
        int i = tid % 12, j = tid / 12;
        if (mode == SLK_STATEK_I) {
            double v = at(2, 2, i, j);
            at(1, 1, i, j) = v; at(1, 2, i, j) = v; at(2, 1, i, j) = v;
            at(0, 2, i, j) = 0.0; at(2, 0, i, j) = 0.0; at(0, 1, i, j) = 0.0; at(1, 0, i, j) = 0.0;
        } else {
            double v = at(1, 1, i, j);
            at(0, 0, i, j) = v; at(0, 1, i, j) = v; at(1, 0, i, j) = v;
        }
    }
    if (tid < 13) {
        if (mode == SLK_STATEK_I) m[13 + tid] = m[26 + tid];
        else m[tid] = m[13 + tid];
    }
}

// Usckf::setMeasurement, Usckf.hpp:322-389 (out of place).  Keeps the 36x36 state block and the
// other feature block's own covariance, sets the new block to R, wipes all cross terms.
__global__ void usckf_set_measurement_kernel(const double *mean, const double *P, double *nmean, double *nP,
                                             const double *z, const double *R, int B, int onfk, int onfkl,
                                             int nfk, int nfkl, int mode, int n)
{
    int oN = 36 + onfk + onfkl, nN = 36 + nfk + nfkl, oNq = oN + 3, nNq = nN + 3;
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per = (long)nN * nN;
    if (e >= per * B) return;
    int b = (int)(e / per), r = (int)((e % per) % nN), c = (int)((e % per) / nN);
    const double *op = P + (size_t)b * oN * oN;
    double v = 0.0;
    if (r < 36 && c < 36) {
        v = op[r + (size_t)c * oN];
    } else if (r >= 36 && c >= 36) {
        bool rk = r < 36 + nfk, ck = c < 36 + nfk;
        if (rk && ck) {
            v = (mode == SLK_STATEK) ? R[(r - 36) + (size_t)(c - 36) * n] : op[r + (size_t)c * oN];
        } else if (!rk && !ck) {
            int i = r - 36 - nfk, j = c - 36 - nfk;
            if (mode == SLK_STATEK_L) v = R[i + (size_t)j * n];
            else {
                // reference reads the kept block at offset DOF + NEW |featuresk| of the OLD matrix (:342)
                int oi = 36 + nfk + i, oj = 36 + nfk + j;
                v = (oi < oN && oj < oN) ? op[oi + (size_t)oj * oN] : 0.0;
            }
        }
    }
    nP[(size_t)b * per + r + (size_t)c * nN] = v;
    if (c == 0) {
        // mean: one thread per (filter, row r) handles storage entry; r runs over nN >= nNq - 3
        const double *om = mean + (size_t)b * oNq;
        double *nm = nmean + (size_t)b * nNq;
        for (int s = r; s < nNq; s += nN) {
            double mv;
            if (s < 39) mv = om[s];
            else if (s < 39 + nfk) mv = (mode == SLK_STATEK) ? z[(size_t)b * n + (s - 39)] : om[s];
            else mv = (mode == SLK_STATEK_L) ? z[(size_t)b * n + (s - 39 - nfk)] : om[39 + onfk + (s - 39 - nfk)];
            nm[s] = mv;
        }
    }
}

} // namespace slk

#include "slk_usckf_fast.hpp"
