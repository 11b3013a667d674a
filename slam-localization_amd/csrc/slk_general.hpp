// slk_general.hpp -- Msckf UKF update with applyDelta (reference src/filters/Msckf.hpp:196-277, :400-431, :499-525, :574-589,
// :659-666, :723-754) for windows of ANY length (the reference's MultiState is unbounded, State.hpp:342, :373-376): the
// shapes beyond the LDS-resident kernels (N > 208).  One 256-thread workgroup per filter, every array in a per-filter
// global workspace, plain loops in the reference's own order of operations -- no matrix cores, no LDS tiling: this is
// the slow path that makes every legal call work, not a tuned one (a few ms per filter at N = 300).
//   L = chol(P) -> Z = h(X) over the implicit sigma points -> mean_z, S, covXZ (with MTK's atan wrap of long rotation
//   columns) -> removeOutliers -> K = covXZ S^-1 (Gauss-Jordan with partial pivoting, like the reference's PartialPivLU)
//   -> P - K S K^T factored afresh -> re-drawn sigma points, manifold mean with the reference's stop rule, one more pass
//   of the deviations against the final mean, P+ = 1/2 sum d d^T.
// Modes as in msckf_step_kernel: emit 2 (sigma points out), 3 (checkSigmaPoints), 4 (innovation + covariance out),
// gate 0 / 1 / 2 (rows chosen by the caller), registered models or Z from the caller (SLK_MODEL_EXTERNAL).
#pragma once
// (included at the end of slk_kernels.hpp: uses its helpers)

namespace slk {

struct GenWs { size_t L, Z, DZ, Cxz, K, Sm, G, zbar, innov, mu, ref, delta, md, wgt, DR, red, total; };
__host__ __device__ inline GenWs general_ws(int N, int Nq, int nso3, int m)
{
    GenWs w;
    const size_t S = 2 * (size_t)N + 1;
    size_t o = 0;
    w.L = o;     o += (size_t)N * N;
    w.Z = o;     o += S * m;
    w.DZ = o;    o += (size_t)N * m;
    w.Cxz = o;   o += (size_t)N * m;
    w.K = o;     o += (size_t)N * m;
    w.Sm = o;    o += (size_t)m * m;
    w.G = o;     o += (size_t)m * (2 * m + 1);
    w.zbar = o;  o += m;
    w.innov = o; o += m;
    w.mu = o;    o += Nq;
    w.ref = o;   o += Nq;
    w.delta = o; o += N;
    w.md = o;    o += N;
    w.wgt = o;   o += (size_t)nso3 * N;            // atan-wrap factor per (block, column)
    w.DR = o;    o += 3 * (size_t)nso3 * S;        // rotation deviations of every sigma point
    w.red = o;   o += 512;
    w.total = (o + 7) & ~(size_t)7;
    return w;
}

// lower Cholesky in place in A (N x N, column-major, lower triangle), left-looking, one workgroup.  Returns -1 or the first
// non-positive pivot (uniform).
__device__ inline int general_cholesky(double *A, int N, int tid, int *flag)
{
    if (tid == 0) *flag = -1;
    __syncthreads();
    for (int j = 0; j < N; ++j) {
        for (int i = j + tid; i < N; i += 256) {        // column j minus the contributions of the columns left of it
            double s = A[i + (size_t)j * N];
            for (int p = 0; p < j; ++p) s -= A[i + (size_t)p * N] * A[j + (size_t)p * N];
            A[i + (size_t)j * N] = s;
        }
        __syncthreads();
        const double d = A[j + (size_t)j * N];
        if (!(d > 0.0)) { if (tid == 0) *flag = j; __syncthreads(); return j; }
        __syncthreads();
        const double sq = sqrt(d);
        for (int i = j + tid; i < N; i += 256) A[i + (size_t)j * N] = (i == j) ? sq : A[i + (size_t)j * N] / sq;
        __syncthreads();
    }
    return -1;
}

// X_i = mu [+] (delta + sgn L(:, j)): component t of the tangent perturbation
__device__ __forceinline__ double gen_pert(const double *L, int N, const double *delta, int t, int i)
{
    const int j = (i > 0) ? ((i - 1) >> 1) : 0;
    const double sgn = (i == 0) ? 0.0 : ((i & 1) ? 1.0 : -1.0);
    const double l = (i > 0 && j <= t) ? sgn * L[t + (size_t)j * N] : 0.0;
    return delta ? delta[t] + l : l;
}
__device__ __forceinline__ Quat gen_sigma_quat(const double *mu, const double *L, int N, const double *delta, int b, int i)
{
    const int to = msckf_toff(b), so = msckf_soff(b);
    return qmul(ldq(mu + so), so3_exp(gen_pert(L, N, delta, to, i), gen_pert(L, N, delta, to + 1, i), gen_pert(L, N, delta, to + 2, i)));
}

__global__ __launch_bounds__(256) void msckf_update_general_kernel(KArgs a)
{
    __shared__ int ish[64];
    __shared__ double sred[256];
    const int bidx = blockIdx.x, tid = threadIdx.x;
    Lay L = a.lay;
    L.kind = SLK_MSCKF;
    const int N = L.N, Nq = L.Nq, m = a.m, nso3 = L.nso3, S = 2 * N + 1;
    const GenWs w = general_ws(N, Nq, nso3, m > 0 ? m : 1);
    double *ws = a.wsL + (size_t)bidx * w.total;
    double *Lm = ws + w.L, *Z = ws + w.Z, *DZ = ws + w.DZ, *Cxz = ws + w.Cxz, *K = ws + w.K, *Sm = ws + w.Sm, *G = ws + w.G;
    double *zbar = ws + w.zbar, *innov = ws + w.innov, *mu = ws + w.mu, *ref = ws + w.ref, *delta = ws + w.delta, *md = ws + w.md;
    double *wgt = ws + w.wgt, *DR = ws + w.DR;
    double *gmean = a.mean + (size_t)bidx * Nq;
    double *gP = a.P + (size_t)bidx * N * N;
    double *omean = a.mean_out ? a.mean_out + (size_t)bidx * Nq : gmean;
    double *oP = a.P_out ? a.P_out + (size_t)bidx * N * N : gP;
    int *idx = ish;                                       // [0..m) surviving rows (m <= MAXM), [40] count, [41] outliers, [44] flag
    int status = 0;
    if (a.do_update && a.emit != 4 && tid == 0) a.outliers[bidx] = 0u;
    for (int e = tid; e < Nq; e += 256) mu[e] = gmean[e];
    for (size_t e = tid; e < (size_t)N * N; e += 256) Lm[e] = gP[e];       // (the factorisation reads the lower triangle only)
    __syncthreads();
    if (!(a.do_update || a.emit >= 2)) return;
    // ---- sigma points of the full state: Msckf.hpp:228-229 -> :400-431
    int fail = general_cholesky(Lm, N, tid, &ish[44]);
    bool redraw = false;
    if (fail >= 0) {
        status |= SLK_ST_LLT_FAIL;
    } else if (a.emit == 3) {                              // checkSigmaPoints (:819-839): re-draw (mu, 0, Pk)
        for (int t = tid; t < N; t += 256) delta[t] = 0.0;
        __syncthreads();
        redraw = true;
    } else if (a.emit == 2) {
        double *X = a.Xout + (size_t)bidx * S * Nq;
        for (size_t e = tid; e < (size_t)S * N; e += 256) {
            const int t = (int)(e % N), i = (int)(e / N);
            int blk = 0, comp = 0;
            const int s = t2s(L, t, blk, comp);
            if (s >= 0) X[(size_t)i * Nq + s] = mu[s] + gen_pert(Lm, N, nullptr, t, i);
        }
        for (size_t e = tid; e < (size_t)S * nso3; e += 256) {
            const int b = (int)(e % nso3), i = (int)(e / nso3);
            const Quat q = gen_sigma_quat(mu, Lm, N, nullptr, b, i);
            stq(X + (size_t)i * Nq + msckf_soff(b), q);
        }
    } else if (!pose_params_ok(a, L, a.mp ? a.mp + (size_t)bidx * a.mp_stride : nullptr)) {
        status |= SLK_ST_BAD_INDEX;
    } else {
        const double *mp = a.mp ? a.mp + (size_t)bidx * a.mp_stride : nullptr;
        // Z = h(X): :231-232
        if (a.mm == SLK_MODEL_EXTERNAL) {
            const double *Ze = a.Zext + (size_t)bidx * S * m;
            for (size_t e = tid; e < (size_t)S * m; e += 256) Z[e] = Ze[e];
        } else {
            const int nf = measure_features(a.mm, m);
            for (size_t e = tid; e < (size_t)S * nf; e += 256) {
                const int f = (int)(e % nf), i = (int)(e / nf);
                double *Zrow = Z + (size_t)i * m;
                if (a.mm == SLK_MM_FEATURE_PROJ) {
                    int tp, sp, b;
                    pose_of(L, (int)mp[4 * f + 3], tp, sp, b);
                    const double px = mu[sp] + gen_pert(Lm, N, nullptr, tp, i), py = mu[sp + 1] + gen_pert(Lm, N, nullptr, tp + 1, i);
                    const double pz = mu[sp + 2] + gen_pert(Lm, N, nullptr, tp + 2, i);
                    const Quat q = gen_sigma_quat(mu, Lm, N, nullptr, b, i);
                    double lx, ly, lz;
                    qrot(qconj(q), mp[4 * f] - px, mp[4 * f + 1] - py, mp[4 * f + 2] - pz, lx, ly, lz);
                    Zrow[2 * f] = lx / lz;
                    Zrow[2 * f + 1] = ly / lz;
                } else {                                   // SLK_MM_POSE_POSITION
                    int tp, sp, b;
                    pose_of(L, (int)mp[0], tp, sp, b);
                    for (int c = 0; c < 3 && c < m; ++c) Zrow[c] = mu[sp + c] + gen_pert(Lm, N, nullptr, tp + c, i);
                }
            }
        }
        // MTK's log uses atan: a rotation column longer than pi wraps, X_i [-] mu = w L(:, j) on that block's rows
        for (size_t e = tid; e < (size_t)nso3 * N; e += 256) {
            const int j = (int)(e % N), b = (int)(e / N), t0 = msckf_toff(b);
            const double v0 = j <= t0 ? Lm[t0 + (size_t)j * N] : 0.0, v1 = j <= t0 + 1 ? Lm[t0 + 1 + (size_t)j * N] : 0.0;
            const double v2 = j <= t0 + 2 ? Lm[t0 + 2 + (size_t)j * N] : 0.0;
            const double th = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
            wgt[e] = (th >= 3.141592653589793) ? 2.0 * atan(tan(0.5 * th)) / th : 1.0;
        }
        __syncthreads();
        // mean_z (:234), innovation (:236)
        for (int r = tid; r < m; r += 256) {
            double s = 0.0;
            for (int i = 0; i < S; ++i) s += Z[(size_t)i * m + r];
            zbar[r] = s / (double)S;
            innov[r] = a.z[(size_t)bidx * m + r] - zbar[r];
        }
        for (size_t e = tid; e < (size_t)N * m; e += 256) {
            const int r = (int)(e % m), j = (int)(e / m);
            DZ[e] = Z[(size_t)(2 * j + 1) * m + r] - Z[(size_t)(2 * j + 2) * m + r];
        }
        __syncthreads();
        // S = cov(Z) + R (:238); covXZ = 1/2 sum (X_i [-] mu)(Z_i - mean_z)^T (:239 -> :635-657): the +- pairs of column j
        // contribute +- w L(:, j) (Z_{2j+1} - Z_{2j+2}), the mean_z terms cancel
        const double *R = a.R + (size_t)bidx * a.r_stride;
        for (int e = tid; e < m * m; e += 256) {
            const int ra = e % m, rb = e / m;
            double s = 0.0;
            for (int i = 0; i < S; ++i) s += (Z[(size_t)i * m + ra] - zbar[ra]) * (Z[(size_t)i * m + rb] - zbar[rb]);
            Sm[e] = 0.5 * s + R[e];
        }
        for (size_t e = tid; e < (size_t)N * m; e += 256) {
            const int t = (int)(e % N), r = (int)(e / N);
            int blk = -1, comp = 0;
            const int s = t2s(L, t, blk, comp);
            double sum = 0.0;
            for (int j = 0; j <= t; ++j) sum += (s < 0 ? wgt[(size_t)blk * N + j] : 1.0) * Lm[t + (size_t)j * N] * DZ[(size_t)j * m + r];
            Cxz[e] = 0.5 * sum;
        }
        __syncthreads();
        // removeOutliers (:241 -> :723-754) incl. the shifted second erase (:741-744)
        if (tid == 0) {
            int cnt = m, i = 0;
            unsigned nout = 0;
            for (int r = 0; r < m; ++r) idx[r] = r;
            if (a.gate == 2) {
                const int *rs = a.rowsel + (size_t)bidx * (m + 2);
                cnt = rs[0] < 0 ? 0 : (rs[0] > m ? m : rs[0]);
                nout = (unsigned)rs[1];
                for (int r = 0; r < cnt; ++r) { const int v = rs[2 + r]; idx[r] = v < 0 ? 0 : (v >= m ? m - 1 : v); }
                i = cnt;
            }
            while (i < cnt / 2) {
                const int p = idx[2 * i], q = idx[2 * i + 1];
                const double s00 = Sm[p + m * p], s01 = Sm[p + m * q], s10 = Sm[q + m * p], s11 = Sm[q + m * q];
                const double det = s00 * s11 - s01 * s10, r0 = innov[p], r1 = innov[q];
                const double d2 = (r0 * (s11 * r0 - s01 * r1) + r1 * (s00 * r1 - s10 * r0)) / det;
                const bool ok = a.gate ? (d2 < 5.99) : true;
                if (!ok) {
                    for (int rep = 0; rep < 2; ++rep) {
                        const int pos = 2 * i + rep, numRows = cnt - 1;
                        if (pos < numRows) for (int q2 = pos; q2 < numRows; ++q2) idx[q2] = idx[q2 + 1];
                        cnt = numRows;
                    }
                    nout++;
                } else {
                    i++;
                }
            }
            ish[40] = cnt;
            ish[41] = (int)nout;
            ish[45] = 0;
        }
        __syncthreads();
        const int mmr = ish[40];
        if (tid == 0 && a.emit != 4) a.outliers[bidx] = (unsigned)ish[41];
        if (a.emit == 4) {
            double *o = a.Xout + (size_t)bidx * (m * m + m);
            for (int e = tid; e < m * m; e += 256) o[e] = Sm[e];
            for (int e = tid; e < m; e += 256) o[m * m + e] = innov[e];
        } else if (mmr == 0) {
            status |= SLK_ST_ALL_REJECTED;
        } else {
            // K = covXZ S^-1 (:257): Gauss-Jordan with partial pivoting on [S | I]
            const int ldj = 2 * mmr + 1;
            for (int e = tid; e < mmr * mmr; e += 256) {
                const int r = e % mmr, c = e / mmr;
                G[r * ldj + c] = Sm[idx[r] + m * idx[c]];
                G[r * ldj + mmr + c] = (r == c) ? 1.0 : 0.0;
            }
            __syncthreads();
            bool singular = false;
            for (int k = 0; k < mmr; ++k) {
                int piv = k;
                double best = fabs(G[k * ldj + k]);
                for (int i = k + 1; i < mmr; ++i) {
                    const double v = fabs(G[i * ldj + k]);
                    if (v > best) { best = v; piv = i; }
                }
                if (!(best > 0.0)) { singular = true; break; }
                __syncthreads();
                if (piv != k)
                    for (int c = tid; c < 2 * mmr; c += 256) { const double t0 = G[k * ldj + c]; G[k * ldj + c] = G[piv * ldj + c]; G[piv * ldj + c] = t0; }
                __syncthreads();
                const double pv = G[k * ldj + k];
                __syncthreads();
                for (int c = tid; c < 2 * mmr; c += 256) G[k * ldj + c] = G[k * ldj + c] / pv;
                __syncthreads();
                for (int r = tid; r < mmr; r += 256) {
                    if (r == k) continue;
                    const double f = G[r * ldj + k];
                    for (int c = 0; c < 2 * mmr; ++c) G[r * ldj + c] -= f * G[k * ldj + c];
                }
                __syncthreads();
            }
            if (singular) {
                status |= SLK_ST_SINGULAR;
            } else {
                for (size_t e = tid; e < (size_t)N * mmr; e += 256) {
                    const int t = (int)(e % N), c = (int)(e / N);
                    double sum = 0.0;
                    for (int c2 = 0; c2 < mmr; ++c2) sum += Cxz[t + (size_t)N * idx[c2]] * G[c2 * ldj + mmr + c];
                    K[e] = sum;
                }
                __syncthreads();
                for (int t = tid; t < N; t += 256) {                     // delta = K * innovation (:263)
                    double sum = 0.0;
                    for (int c = 0; c < mmr; ++c) sum += K[t + (size_t)N * c] * innov[idx[c]];
                    delta[t] = sum;
                }
                // Pk -= K S K^T (:262; K S = covXZ), lower triangle, then its factor for applyDelta (:659-662)
                for (size_t e = tid; e < (size_t)N * N; e += 256) {
                    const int i = (int)(e % N), j = (int)(e / N);
                    if (i < j) continue;
                    double sum = 0.0;
                    for (int c = 0; c < mmr; ++c) sum += Cxz[i + (size_t)N * idx[c]] * K[j + (size_t)N * c];
                    Lm[e] = gP[e] - sum;
                }
                __syncthreads();
                fail = general_cholesky(Lm, N, tid, &ish[44]);
                if (fail >= 0) status |= SLK_ST_LLT_FAIL;
                else redraw = true;
            }
        }
    }
    if (redraw) {
        // ---- re-drawn sigma points, manifold mean (:664 -> :499-525), covariance (:665 -> :574-589)
        for (int t = tid; t < N; t += 256) {
            int blk = 0, comp = 0;
            const int s = t2s(L, t, blk, comp);
            if (s >= 0) ref[s] = mu[s] + delta[t];
        }
        for (int b = tid; b < nso3; b += 256) stq(ref + msckf_soff(b), gen_sigma_quat(mu, Lm, N, delta, b, 0));
        __syncthreads();
        int it = 0;
        bool final_pass = false;
        for (;;) {
            for (size_t e = tid; e < (size_t)nso3 * S; e += 256) {       // rotation blocks of X_i [-] ref
                const int b = (int)(e % nso3), i = (int)(e / nso3);
                double dx, dy, dz;
                so3_boxminus(gen_sigma_quat(mu, Lm, N, delta, b, i), ldq(ref + msckf_soff(b)), dx, dy, dz);
                double *o = DR + 3 * ((size_t)b * S + i);
                o[0] = dx; o[1] = dy; o[2] = dz;
            }
            __syncthreads();
            if (final_pass) break;
            // mean_delta = sum_i (X_i [-] ref) / S (:507-509); vector rows: the +- L terms cancel
            for (int t = tid; t < N; t += 256) {
                int blk = 0, comp = 0;
                const int s = t2s(L, t, blk, comp);
                if (s >= 0) {
                    md[t] = (mu[s] + delta[t]) - ref[s];
                } else {
                    double sum = 0.0;
                    const double *row = DR + 3 * (size_t)blk * S + comp;
                    for (int i = 0; i < S; ++i) sum += row[3 * (size_t)i];
                    md[t] = sum / (double)S;
                }
            }
            __syncthreads();
            double n2 = 0.0;
            for (int t = tid; t < N; t += 256) n2 += md[t] * md[t];
            sred[tid] = n2;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) { if (tid < o) sred[tid] += sred[tid + o]; __syncthreads(); }
            const double norm = sqrt(sred[0]);
            __syncthreads();
            for (int t = tid; t < N; t += 256) {                         // reference += mean_delta (:510)
                int blk = 0, comp = 0;
                const int s = t2s(L, t, blk, comp);
                if (s >= 0) ref[s] = ref[s] + md[t];
            }
            for (int b = tid; b < nso3; b += 256) {
                const int to = msckf_toff(b), so = msckf_soff(b);
                stq(ref + so, qmul(ldq(ref + so), so3_exp(md[to], md[to + 1], md[to + 2])));
            }
            __syncthreads();
            if (!(norm > 1e-6 && ++it < 10000)) final_pass = true;       // :511, then the deviations against the final mean
        }
        if (it >= 10000) status |= SLK_ST_MEAN_NOT_CONVERGED;
        for (int e = tid; e < Nq; e += 256) omean[e] = ref[e];
        // P+ = 1/2 sum_i d_i d_i^T (:574-589), d_i = X_i [-] mean
        auto dev = [&](int t, int s, int blk, int comp, int i) -> double {
            if (s >= 0) return (mu[s] + gen_pert(Lm, N, delta, t, i)) - ref[s];
            return DR[3 * ((size_t)blk * S + i) + comp];
        };
        for (size_t e = tid; e < (size_t)N * N; e += 256) {
            const int r = (int)(e % N), c = (int)(e / N);
            if (r < c) continue;
            int br = 0, cr = 0, bc = 0, cc = 0;
            const int sr = t2s(L, r, br, cr), sc = t2s(L, c, bc, cc);
            double sum = 0.0;
            for (int i = 0; i < S; ++i) sum += dev(r, sr, br, cr, i) * dev(c, sc, bc, cc, i);
            oP[r + (size_t)c * N] = 0.5 * sum;
            oP[c + (size_t)r * N] = 0.5 * sum;
        }
    }
    if (tid == 0 && status) atomicOr(a.status + bidx, status);
}

} // namespace slk
