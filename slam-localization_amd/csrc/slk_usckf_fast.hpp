// slk_usckf_fast.hpp -- exact-shape fast path of Usckf::update (reference src/filters/Usckf.hpp:246-308, sigma points :532-561,
// moments :630-737) for the unit-test layout (test/UsckfUnitTest.cpp: statek, statek_l, statek_i, 3 + 9 features: N = 48;
// the relative-transform measurement model :62-86, m = 3 rows), 128 threads = two waves per filter, the packed factor in
// LDS (factored by wave 0 at the top of the kernel -- cholp_factor -- or handed over by msckf_chol_kernel), the covariance in global memory.
//   * wave 0 evaluates h at the "+" sigma point of column j = lane, wave 1 at the "-" one (lanes >= 48 evaluate X_0: Z_0 by
//     v_readlane), both SO(3) exponentials of a point in lockstep with the series coefficients in an LDS table;
//   * mean_z, innovation and S = 1/2 sum (Z - mean_z)(Z - mean_z)^T + R from wave reductions of the deviations about Z_0;
//   * covXZ = 1/2 L dZ as 3 x 48 multiply-adds per lane (m = 3: the matrix cores would idle 13 of 16 columns), the column
//     range split over the two waves; K = covXZ S^-1 through the 3 x 3 Cholesky factor every lane holds in registers;
//   * Pk -= covXZ K^T as a read-modify-write of global memory with all eighteen loads of a thread in flight; mu [+] K nu.
// Returns false -- before its first global write -- for a failed factorisation, a rotation column that may exceed pi, a
// non-SPD S, other models / modes: the general kernel body runs instead.
#pragma once
// (included at the end of slk_usckf.hpp)

namespace slk {

struct UFast {
    static constexpr int N = 48, Nq = 51, S = 97;
    static constexpr int oL = 0;                  // packed factor, 1176
    static constexpr int oMu = 1176;              // 52
    static constexpr int oT = oMu + 52;           // series table, 26
    static constexpr int oYp = oT + 26;           // deviations of Z about Z_0, "+" points [64][4]
    static constexpr int oYm = oYp + 256;         // "-" points
    static constexpr int oRed = oYm + 256;        // reductions of the two waves: 2 x 16
    static constexpr int oPx = oRed + 32;         // covXZ partial sums of the two waves [2][48][4], then covXZ [48][4]
    static constexpr int oK = oPx + 384;          // K [48][4] (K nu in [.][3])
    static constexpr int oInts = oK + 192;        // flags
    static constexpr int total = oInts + 8;
};

__device__ __forceinline__ bool usckf_update_fast(const KArgs &a, double *smem)
{
    using F = UFast;
    constexpr int N = F::N, Nq = F::Nq, S = F::S;
    if (a.mm != SLK_MM_VO_RELATIVE || a.emit != 0 || a.m != 3 || !a.do_update || a.do_predict || a.gate > 9) return false;
    const int bidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *Lp = smem + F::oL, *mu = smem + F::oMu, *T = smem + F::oT, *Yp = smem + F::oYp, *Ym = smem + F::oYm, *red = smem + F::oRed;
    double *Px = smem + F::oPx, *Kg = smem + F::oK;
    int *flags = reinterpret_cast<int *>(smem + F::oInts);
    double *gmean = a.mean + (size_t)bidx * Nq;
    double *gP = a.P + (size_t)bidx * N * N;
    // ---- load: factor, mean, series table; the conditions that hand the filter to the general body meet in one barrier
    {
        // the factor: from msckf_chol_kernel's workspace, or (a.wsfail == nullptr) factored HERE by wave 0 -- panel by rows,
        // straight into LDS, no round trip through memory (Usckf.hpp:532-538); the colbuf of the factorisation is the Yp array
        const bool here = a.wsfail == nullptr;
        const double *gL = here ? gP : a.wsL + (size_t)bidx * pk_size(N);
        double v[10];
        int bad = 0;
        if (!here) {
#pragma unroll
            for (int q = 0; q < 10; ++q) { const int e = tid + 128 * q; v[q] = gL[e < pk_size(N) ? e : 0]; }
            bad = a.wsfail[bidx] >= 0;
        } else if (wave == 0) {
            d4 acc[CholM<3>::NTL];
            cholm_load_t<3>(acc, N, lane, [&](int i, int j) { return gP[i + (size_t)j * N]; });
            bad = cholp_factor<3, 0>(acc, Lp, N, Yp, lane) >= 0;
        }
        const double m0 = (tid < Nq) ? gmean[tid] : 0.0;
        const double tv = fast_series_table[tid < 26 ? tid : 25];
        if (tid < 3) {                                          // rotation rows 3..5 of the three single states (Usckf.hpp:553-556)
            const int t0 = 12 * tid + 3;
            const double sd = gP[t0 * (N + 1)] + gP[(t0 + 1) * (N + 1)] + gP[(t0 + 2) * (N + 1)];
            bad |= !(sd < 9.869604401089358);
        }
        if (!here) {
#pragma unroll
            for (int q = 0; q < 10; ++q) { const int e = tid + 128 * q; if (e < pk_size(N)) Lp[e] = v[q]; }
        }
        if (tid < Nq) mu[tid] = m0;
        if (tid < 26) T[tid] = tv;
        if (tid < 2) flags[tid] = 0;
        if (__syncthreads_or(bad)) return false;
    }
    // ---- Z = h(X) (Usckf.hpp:277-278): wave 0 the "+" point of column lane, wave 1 the "-" point; lanes >= 48: X_0
    double y[3];
    {
        const int j = lane;
        const bool act = j < N;
        const double sg = wave ? -1.0 : 1.0;
        const int jb = pkcol(N, act ? j : 0);
        auto lzz = [&](int t) __attribute__((always_inline)) -> double { const double v = Lp[(act && j <= t) ? jb + t : 0]; return (act && j <= t) ? sg * v : 0.0; };
        const double pk0 = mu[0] + lzz(0), pk1 = mu[1] + lzz(1), pk2 = mu[2] + lzz(2);
        const double pi0 = mu[26] + lzz(24), pi1 = mu[27] + lzz(25), pi2 = mu[28] + lzz(26);
        const double f0 = mu[39] + lzz(36), f1 = mu[40] + lzz(37), f2 = mu[41] + lzz(38);
        const double rv[2][3] = {{lzz(3), lzz(4), lzz(5)}, {lzz(27), lzz(28), lzz(29)}};
        Quat ex[2];
        bool ok = so3_exp_tab<2>(T, rv, ex);
        const Quat qk = qmul(ldq(mu + 3), ex[0]), qi = qmul(ldq(mu + 29), ex[1]);
        const Quat rel[1] = {qmul(qconj(qi), qk)};               // delta_state = statek - statek_i (UsckfUnitTest.cpp:66)
        double r3[1][3];
        so3_log_tab<1>(T, rel, r3);
        Quat dq[1];
        ok = so3_exp_tab<1>(T, r3, dq) && ok;                   // ... assigned to a WSingleState: set()
        if (!__all(ok)) flags[1] = 1;
        double o0, o1, o2;
        qmat_apply(dq[0], f0, f1, f2, o0, o1, o2);
        const double z0 = o0 + (pk0 - pi0), z1 = o1 + (pk1 - pi1), z2 = o2 + (pk2 - pi2);
        const double Z0 = readlane_f64(z0, 63), Z1 = readlane_f64(z1, 63), Z2 = readlane_f64(z2, 63);
        y[0] = z0 - Z0; y[1] = z1 - Z1; y[2] = z2 - Z2;         // (lanes >= 48: exactly zero)
        double *Yw = wave ? Ym : Yp;
        Yw[4 * lane] = y[0]; Yw[4 * lane + 1] = y[1]; Yw[4 * lane + 2] = y[2]; Yw[4 * lane + 3] = 0.0;
        // sums over the columns: y (3) and y y^T (6)
        double *rw = red + 16 * wave;
        const double s0 = wave_sum_f64(y[0]), s1 = wave_sum_f64(y[1]), s2 = wave_sum_f64(y[2]);
        const double q00 = wave_sum_f64(y[0] * y[0]), q10 = wave_sum_f64(y[1] * y[0]), q11 = wave_sum_f64(y[1] * y[1]);
        const double q20 = wave_sum_f64(y[2] * y[0]), q21 = wave_sum_f64(y[2] * y[1]), q22 = wave_sum_f64(y[2] * y[2]);
        if (lane == 0) {
            rw[0] = s0; rw[1] = s1; rw[2] = s2; rw[3] = q00; rw[4] = q10; rw[5] = q11; rw[6] = q20; rw[7] = q21; rw[8] = q22;
            rw[9] = Z0; rw[10] = Z1; rw[11] = Z2;
        }
    }
    __syncthreads();
    if (flags[1]) return false;                                  // a rotation beyond the series' domain: the general body
    // ---- mean_z (:279), innovation (:281), S = cov(Z) + R (:283-284) -- every lane the same 3 x 3; its Cholesky factor
    double nu[3], g00, g10, g11, g20, g21, g22;                  // G lower; the diagonal as reciprocals
    {
        const double *R = a.R + (size_t)bidx * a.r_stride;
        const double d0 = (red[0] + red[16]) / (double)S, d1 = (red[1] + red[17]) / (double)S, d2 = (red[2] + red[18]) / (double)S;
        nu[0] = a.z[(size_t)bidx * 3] - (red[9] + d0);
        nu[1] = a.z[(size_t)bidx * 3 + 1] - (red[10] + d1);
        nu[2] = a.z[(size_t)bidx * 3 + 2] - (red[11] + d2);
        const double s00 = 0.5 * ((red[3] + red[19]) - (double)S * d0 * d0) + R[0];
        const double s10 = 0.5 * ((red[4] + red[20]) - (double)S * d1 * d0) + R[1];
        const double s11 = 0.5 * ((red[5] + red[21]) - (double)S * d1 * d1) + R[4];
        const double s20 = 0.5 * ((red[6] + red[22]) - (double)S * d2 * d0) + R[2];
        const double s21 = 0.5 * ((red[7] + red[23]) - (double)S * d2 * d1) + R[5];
        const double s22 = 0.5 * ((red[8] + red[24]) - (double)S * d2 * d2) + R[8];
        bool spd = s00 > 0.0;
        double sq, rs;
        rsqrt_pivot(s00, sq, rs);
        g00 = rs; g10 = s10 * rs; g20 = s20 * rs;
        double dd = fma(-g10, g10, s11);
        spd = spd && dd > 0.0;
        rsqrt_pivot(dd, sq, rs);
        g11 = rs; g21 = fma(-g20, g10, s21) * rs;
        dd = fma(-g21, g21, fma(-g20, g20, s22));
        spd = spd && dd > 0.0;
        rsqrt_pivot(dd, sq, rs);
        g22 = rs;
        if (!spd) return false;                                  // (uniform) non-SPD S: SLK_ST_SINGULAR by the general body
    }
    // ---- covXZ = 1/2 L dZ (:283 -> :714-737): lane t = row t, wave w the columns 24 w .. 24 w + 23
    {
        const int t = lane;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
#pragma unroll
        for (int jj = 0; jj < 24; ++jj) {
            const int j = 24 * wave + jj;
            const double dz0 = Yp[4 * j] - Ym[4 * j], dz1 = Yp[4 * j + 1] - Ym[4 * j + 1], dz2 = Yp[4 * j + 2] - Ym[4 * j + 2];
            if (t < N && t >= j) {
                const double l = Lp[pkcol(N, j) + t];
                c0 = fma(l, dz0, c0); c1 = fma(l, dz1, c1); c2 = fma(l, dz2, c2);
            }
        }
        if (t < N) { double *o = Px + 192 * wave + 4 * t; o[0] = 0.5 * c0; o[1] = 0.5 * c1; o[2] = 0.5 * c2; }
    }
    __syncthreads();
    // ---- K = covXZ S^-1 (:286-288), mahalanobis (:292), K nu; rows on the lanes of wave 0
    double w0, w1, w2;                                           // Ls^-1 nu
    w0 = nu[0] * g00; w1 = fma(-g10, w0, nu[1]) * g11; w2 = fma(-g21, w1, fma(-g20, w0, nu[2])) * g22;
    bool accept = true;
    if (a.gate > 0) {
        const double d2m = w0 * w0 + w1 * w1 + w2 * w2;
        const double thr = a.gate == 1 ? 3.84 : a.gate == 2 ? 5.99 : a.gate == 3 ? 7.81 : a.gate == 4 ? 9.49 : a.gate == 5 ? 11.07
                         : a.gate == 6 ? 12.59 : a.gate == 7 ? 14.07 : a.gate == 8 ? 15.51 : 16.92;      // Usckf.hpp:794-855
        accept = d2m < thr;
    }
    if (!accept) {                                               // (uniform) :293-294: nothing is applied
        if (tid == 0) { a.outliers[bidx] = 1u; atomicOr(a.status + bidx, SLK_ST_ALL_REJECTED); }
        return true;
    }
    if (wave == 0 && lane < N) {
        const int t = lane;
        const double p0 = Px[4 * t] + Px[192 + 4 * t], p1 = Px[4 * t + 1] + Px[192 + 4 * t + 1], p2 = Px[4 * t + 2] + Px[192 + 4 * t + 2];
        double k0 = p0 * g00, k1 = fma(-g10, k0, p1) * g11, k2 = fma(-g21, k1, fma(-g20, k0, p2)) * g22;      // forward: Ls w = p
        k2 = k2 * g22; k1 = fma(-g21, k2, k1) * g11; k0 = fma(-g10, k1, fma(-g20, k2, k0)) * g00;             // backward: Ls^T x = w
        Px[4 * t] = p0; Px[4 * t + 1] = p1; Px[4 * t + 2] = p2;
        Kg[4 * t] = k0; Kg[4 * t + 1] = k1; Kg[4 * t + 2] = k2;
        Kg[4 * t + 3] = k0 * nu[0] + k1 * nu[1] + k2 * nu[2];   // delta = K nu (:299)
    }
    __syncthreads();
    // ---- Pk -= K S K^T (:296; K S = covXZ): read-modify-write of the covariance in global memory, its loads first
    if (a.lower_only) {
        // the six lower 16 x 16 tiles only (the diagonal ones whole): nothing on the device reads the strict upper triangle
        // (Usckf.hpp:537: Eigen::LLT), the host completes it when somebody wants the matrix (slk_mirror_upper_kernel)
        const int r = tid & 15, c0 = tid >> 4;
        double pv[12];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int I = t < 1 ? 0 : (t < 3 ? 1 : 2), J = t - I * (I + 1) / 2;
#pragma unroll
            for (int h = 0; h < 2; ++h) pv[2 * t + h] = gP[(16 * I + r) + (size_t)(16 * J + c0 + 8 * h) * N];
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int I = t < 1 ? 0 : (t < 3 ? 1 : 2), J = t - I * (I + 1) / 2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 16 * I + r, jc = 16 * J + c0 + 8 * h;
                const double s = Px[4 * i] * Kg[4 * jc] + Px[4 * i + 1] * Kg[4 * jc + 1] + Px[4 * i + 2] * Kg[4 * jc + 2];
                gP[i + (size_t)jc * N] = pv[2 * t + h] - s;
            }
        }
    } else {
        double pv[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) pv[q] = gP[tid + 128 * q];
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const int e = tid + 128 * q, i = e % N, jc = e / N;
            const double s = Px[4 * i] * Kg[4 * jc] + Px[4 * i + 1] * Kg[4 * jc + 1] + Px[4 * i + 2] * Kg[4 * jc + 2];
            gP[e] = pv[q] - s;
        }
    }
    // ---- mu_state = mu_state + state(K nu) (:299-301)
    if (wave == 1) {
        const int t = lane;
        if (t < N) {
            int blk = 0, comp = 0;
            Lay L = a.lay;
            L.kind = SLK_USCKF; L.nfk = 3; L.nfkl = 9; L.N = 48; L.Nq = 51; L.nso3 = 3;
            const int s = t2s(L, t, blk, comp);
            if (s >= 0) gmean[s] = mu[s] + Kg[4 * t + 3];
        }
        if (lane >= 48 && lane < 51) {
            const int b = lane - 48, to = 12 * b + 3, so = 13 * b + 3;
            const double dv[1][3] = {{Kg[4 * to + 3], Kg[4 * (to + 1) + 3], Kg[4 * (to + 2) + 3]}};
            Quat ex[1];
            Quat qn;
            if (so3_exp_tab<1>(T, dv, ex)) qn = qmul(ldq(mu + so), ex[0]);
            else qn = qmul(ldq(mu + so), so3_exp(dv[0][0], dv[0][1], dv[0][2]));
            gmean[so] = qn.x; gmean[so + 1] = qn.y; gmean[so + 2] = qn.z; gmean[so + 3] = qn.w;
        }
    }
    if (tid == 0) a.outliers[bidx] = 0u;
    return true;
}

} // namespace slk
