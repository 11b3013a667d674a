// slk_usckf.hpp -- HIP kernels for localization::Usckf (reference src/filters/Usckf.hpp):
// predict with clone / feature cross-covariance propagation (:107-244), UKF update with direct
// boxplus correction (:246-308), cloning (:391-433) and setMeasurement (:322-389).
// One workgroup per filter, covariance resident in LDS.
#pragma once
#include "slk_kernels.hpp"

namespace slk {

struct UCarve { int P, Lm, mu, small, pool, total; int lda, S; };

__host__ __device__ inline UCarve carve_usckf(int N, int Nq, int m)
{
    UCarve c;
    c.lda = N | 1;
    c.S = 2 * N + 1;
    int o = 0;
    c.P = o;     o += round_up(N * c.lda, 2);
    c.Lm = o;    o += round_up(N * c.lda, 2);
    c.mu = o;    o += round_up(Nq, 2);
    c.small = o; o += 64;
    c.pool = o;
    int upd = round_up(c.S * m, 2) + 3 * round_up(N * m, 2) + round_up(m * m, 2) + round_up(m * (2 * m + 1), 2)
              + 4 * round_up(m, 2) + round_up(N, 2);
    int pred = PRED_SCRATCH + 160 + 160 + 2 * round_up(12 * N, 2);
    c.total = o + (upd > pred ? upd : pred);
    return c;
}

template <int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void usckf_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int bidx = blockIdx.x, tid = threadIdx.x;
    const Lay L = a.lay;
    const int N = L.N, Nq = L.Nq, m = a.m;
    const UCarve cv = carve_usckf(N, Nq, m);
    const int lda = cv.lda, S = cv.S;
    double *P = smem + cv.P, *Lm = smem + cv.Lm, *mu = smem + cv.mu, *pool = smem + cv.pool;
    int *ish = reinterpret_cast<int *>(smem + cv.small);
    double *gmean = a.mean + (size_t)bidx * Nq;
    double *gP = a.P + (size_t)bidx * N * N;
    int status = 0;
    if (a.do_update && tid == 0) a.outliers[bidx] = 0u;

    for (int e = tid; e < Nq; e += NTHREADS) mu[e] = gmean[e];
    for (int e = tid; e < N * N; e += NTHREADS) { int r = e % N, c = e / N; P[r + c * lda] = gP[e]; }
    __syncthreads();

    if (a.do_predict || a.emit == 1) {
        // ---- Usckf::predict, Usckf.hpp:107-244
        double *Pblk = pool, *Pn = pool + 160, *Pxy = pool + 320, *Fk = pool + 480, *scr = pool + 640;
        double *RB = pool + 640 + 672;             // old rows 24..35 of P: 12 x N (ld 12)
        double *CB = RB + round_up(12 * N, 2);     // old cols 24..35 of P: N x 12 (ld N)
        for (int e = tid; e < 144; e += NTHREADS) { int r = e % 12, c = e / 12; Pblk[r + c * 13] = P[(24 + r) + (24 + c) * lda]; }
        __syncthreads();
        int st = predict_phase<NTHREADS, true>(a, bidx, tid, Pblk, mu + 26, Pn, scr, Pxy);
        if (a.emit == 1) return;
        status |= st;
        if (!(st & SLK_ST_LLT_FAIL)) {
            // Fk = Pxy^T * Pk_i^-1 (:154).  Pk_i = L L^T (its Cholesky factor is in Pblk), so
            // Fk^T = Pk_i^-1 Pxy: one forward + one backward substitution per column.
            if (tid < 12) {
                double x[12];
                for (int r = 0; r < 12; ++r) {
                    double s = Pxy[r + 12 * tid];
                    for (int p = 0; p < r; ++p) s -= Pblk[r + p * 13] * x[p];
                    x[r] = s / Pblk[r + r * 13];
                }
                for (int r = 11; r >= 0; --r) {
                    double s = x[r];
                    for (int p = r + 1; p < 12; ++p) s -= Pblk[p + r * 13] * x[p];
                    x[r] = s / Pblk[r + r * 13];
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
                for (int e = tid; e < N * N; e += NTHREADS) { int r = e % N, c = e / N; gP[e] = P[r + c * lda]; }
                for (int e = tid; e < Nq; e += NTHREADS) gmean[e] = mu[e];
            }
        }
        __syncthreads();
    }

    if (a.do_update || a.emit == 2) {
        // ---- Usckf::update, Usckf.hpp:246-308
        for (int e = tid; e < N * N; e += NTHREADS) { int r = e % N, c = e / N; Lm[r + c * lda] = P[r + c * lda]; }
        __syncthreads();
        int fail = chol_lower_inplace<NTHREADS>(Lm, N, lda, tid);
        bool applied = false;
        if (fail >= 0) {
            status |= SLK_ST_LLT_FAIL;
        } else if (a.emit == 2) {
            double *X = a.Xout + (size_t)bidx * S * Nq;
            for (int e = tid; e < S * N; e += NTHREADS) {
                int t = e % N, i = e / N, blk = 0, comp = 0;
                int s = t2s(L, t, blk, comp);
                if (s >= 0) X[(size_t)i * Nq + s] = mu[s] + pert(Lm, lda, nullptr, t, sig_of(i));
            }
            for (int e = tid; e < S * 3; e += NTHREADS) {
                int b = e % 3, i = e / 3;
                Quat q = sigma_quat(L, mu, Lm, lda, nullptr, b, sig_of(i));
                double *o = X + (size_t)i * Nq + so3_soff(L, b);
                o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
            }
        } else {
            double *Z = pool;
            double *Pxz = Z + round_up(S * m, 2);
            double *K = Pxz + round_up(N * m, 2);
            double *KS = K + round_up(N * m, 2);
            double *Sm = KS + round_up(N * m, 2);
            double *G = Sm + round_up(m * m, 2);
            double *zbar = G + round_up(m * (2 * m + 1), 2);
            double *innov = zbar + round_up(m, 2);
            double *dlt = innov + 2 * round_up(m, 2);
            const double *mp = a.mp ? a.mp + (size_t)bidx * a.mp_stride : nullptr;
            if (a.mm == SLK_MODEL_EXTERNAL) {
                const double *Ze = a.Zext + (size_t)bidx * S * m;
                for (int e = tid; e < S * m; e += NTHREADS) Z[e] = Ze[e];
            } else {
                int nf = measure_features(a.mm, m);
                for (int e = tid; e < S * nf; e += NTHREADS) {
                    int f = e % nf, i = e / nf;
                    measure_item(a, L, mp, mu, Lm, lda, i, f, Z + i * m);
                }
            }
            __syncthreads();
            for (int r = tid; r < m; r += NTHREADS) {                   // meanZ :280, innovation :290
                double sum = 0.0;
                for (int i = 0; i < S; ++i) sum += Z[i * m + r];
                double zb = sum / (double)S;
                zbar[r] = zb;
                innov[r] = a.z[(size_t)bidx * m + r] - zb;
            }
            __syncthreads();
            const double *R = a.R + (size_t)bidx * a.r_stride;
            for (int e = tid; e < m * m; e += NTHREADS) {               // S :282
                int r = e % m, c = e / m;
                double zr = zbar[r], zc = zbar[c], sum = 0.0;
                for (int i = 0; i < S; ++i) sum += (Z[i * m + r] - zr) * (Z[i * m + c] - zc);
                Sm[e] = 0.5 * sum + R[e];
            }
            for (int e = tid; e < N * m; e += NTHREADS) {               // covXZ :283 -> :714-737
                int t = e % N, r = e / N, blk = -1, comp = 0;
                int s = t2s(L, t, blk, comp);
                double sum = 0.0;
                if (s >= 0) {
                    for (int j = 0; j <= t; ++j) sum += Lm[t + j * lda] * (Z[(2 * j + 1) * m + r] - Z[(2 * j + 2) * m + r]);
                } else {
                    int t0 = t - comp;
                    for (int j = 0; j <= t; ++j) {
                        double v0 = Lz(Lm, lda, t0, j), v1 = Lz(Lm, lda, t0 + 1, j), v2 = Lz(Lm, lda, t0 + 2, j);
                        double th = sqrt(v0 * v0 + v1 * v1 + v2 * v2), w = 1.0;
                        if (th >= 3.141592653589793) w = 2.0 * atan(tan(0.5 * th)) / th;
                        sum += w * Lm[t + j * lda] * (Z[(2 * j + 1) * m + r] - Z[(2 * j + 2) * m + r]);
                    }
                }
                Pxz[e] = 0.5 * sum;
            }
            __syncthreads();
            // S^-1 (:285-286) by Gauss-Jordan with partial pivoting
            const int ldg = 2 * m + 1;
            for (int e = tid; e < m * m; e += NTHREADS) {
                int r = e % m, c = e / m;
                G[r * ldg + c] = Sm[r + m * c];
                G[r * ldg + m + c] = (r == c) ? 1.0 : 0.0;
            }
            __syncthreads();
            bool singular = false;
            for (int k = 0; k < m; ++k) {
                int piv = k;
                double best = fabs(G[k * ldg + k]);
                for (int i = k + 1; i < m; ++i) {
                    double v = fabs(G[i * ldg + k]);
                    if (v > best) { best = v; piv = i; }
                }
                if (!(best > 0.0)) { singular = true; break; }
                __syncthreads();
                if (piv != k)
                    for (int c = tid; c < 2 * m; c += NTHREADS) {
                        double t0 = G[k * ldg + c]; G[k * ldg + c] = G[piv * ldg + c]; G[piv * ldg + c] = t0;
                    }
                __syncthreads();
                double pv = G[k * ldg + k];
                __syncthreads();
                for (int c = tid; c < 2 * m; c += NTHREADS) G[k * ldg + c] = G[k * ldg + c] / pv;
                __syncthreads();
                for (int r = tid; r < m; r += NTHREADS) {
                    if (r == k) continue;
                    double f = G[r * ldg + k];
                    for (int c = 0; c < 2 * m; ++c) G[r * ldg + c] -= f * G[k * ldg + c];
                }
                __syncthreads();
            }
            if (singular) {
                status |= SLK_ST_SINGULAR;
            } else {
                for (int e = tid; e < N * m; e += NTHREADS) {           // K = covXZ * S^-1 :288
                    int t = e % N, c = e / N;
                    double sum = 0.0;
                    for (int c2 = 0; c2 < m; ++c2) sum += Pxz[t + N * c2] * G[c2 * ldg + m + c];
                    K[e] = sum;
                }
                double d2 = 0.0;                                        // mahalanobis2 :292
                for (int i = 0; i < m; ++i) {
                    double s = 0.0;
                    for (int j = 0; j < m; ++j) s += G[i * ldg + m + j] * innov[j];
                    d2 += innov[i] * s;
                }
                bool ok = true;
                if (a.gate > 0) {
                    const double thr[10] = {0, 3.84, 5.99, 7.81, 9.49, 11.07, 12.59, 14.07, 15.51, 16.92};
                    ok = (a.gate <= 9) ? (d2 < thr[a.gate]) : false;   // Usckf.hpp:794-855
                }
                __syncthreads();
                if (!ok) {
                    if (tid == 0) a.outliers[bidx] = 1u;
                    status |= SLK_ST_ALL_REJECTED;
                } else {
                    for (int e = tid; e < N * m; e += NTHREADS) {
                        int t = e % N, c = e / N;
                        double sum = 0.0;
                        for (int c2 = 0; c2 < m; ++c2) sum += K[t + N * c2] * Sm[c2 + m * c];
                        KS[e] = sum;
                    }
                    for (int t = tid; t < N; t += NTHREADS) {
                        double sum = 0.0;
                        for (int c = 0; c < m; ++c) sum += K[t + N * c] * innov[c];
                        dlt[t] = sum;
                    }
                    __syncthreads();
                    for (int e = tid; e < N * N; e += NTHREADS) {       // Pk -= K S K^T :296
                        int i = e % N, j = e / N;
                        double sum = 0.0;
                        for (int c = 0; c < m; ++c) sum += KS[i + N * c] * K[j + N * c];
                        P[i + j * lda] -= sum;
                    }
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
        if (a.emit != 2 && (applied || a.do_predict)) {
            for (int e = tid; e < N * N; e += NTHREADS) { int r = e % N, c = e / N; gP[e] = P[r + c * lda]; }
            for (int e = tid; e < Nq; e += NTHREADS) gmean[e] = mu[e];
        }
    }
    (void)ish;
    if (tid == 0 && status) atomicOr(a.status + bidx, status);
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
