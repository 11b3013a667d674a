// slk_step_fast.hpp -- exact-shape fast path of the Msckf UKF update with applyDelta (reference src/filters/Msckf.hpp:196-277,
// :400-431, :499-525, :574-589, :659-666, :723-754) for k = 4 .. 8 clones (N = 36 .. 60), m = 8 rows = four 2-D features
// of the registered feature-projection model, 256 threads = four waves per filter.
//
// Same algorithm as msckf_step_kernel's body (slk_kernels.hpp) -- implicit sigma points, S / gate / gain, applyDelta's
// factor as a factor update L' = L chol(I - B B^T), manifold mean with the reference's stop rule, odd / even covariance
// rebuild -- re-laid out so that the hot loops carry no index arithmetic:
//   * the factor lives in LDS as 16 x 16 TILES (tile (I, J) of the lower triangle at (I (I + 1) / 2 + J) * 256, element
//     (t, j) at (j & 15) * 16 + (t & 15), exact zeros above the diagonal and in the padding): every MFMA operand fetch of
//     the factor -- factor update, covariance rebuild -- is `ds_read_b64 base(lane) offset:imm`, conflict free;
//   * one wave per feature evaluates h(X) for the +- pair of a column and the centre point at once (Z_0 by v_readlane, no
//     barrier), S comes from 15 MFMAs on [y+ ; y-] rows, delta = K nu is ONE matrix-vector product L (1/2 dZ S^-1 nu) -- the
//     gain K and covXZ are never formed;
//   * the mean loop works on (block, column) PAIRS: both deviations stay in registers across the convergence test, the
//     first-order correction against the final mean (see slk_kernels.hpp) is applied there, the odd parts go straight into
//     the rotation rows of the tiled factor (so that O = the factor array) and the even parts minus the centre deviation
//     into a k-step-major array E^ in the index space of the rotation rows:
//         P+ = O O^T + E E^T + 1/2 d0 d0^T = O O^T + E^ E^^T + p d0^T + d0 p^T,   p = sum_j e^_j + (N + 1/2) / 2 d0,
//     O and E^ are zero beyond column t + 2 of row t: tile column J needs the k-steps 0 .. 4 J + 3 only (79 + 83 MFMAs
//     instead of 150 + 48 with gathers), one wave per tile column, no cross-wave reduction;
//   * tiles leave from the accumulators (mirror triangle) and through a private LDS transpose (lower triangle).
// Anything rare -- failed factorisation, rotation column beyond pi, non-SPD innovation covariance, every block gated out,
// an indefinite downdate, a mean that needs more than 64 rounds, other models / gate modes -- returns false BEFORE the
// first global write and the general body runs instead.
#pragma once
// (included at the end of slk_kernels.hpp: uses its helpers)
#ifdef SLK_STAMPS
#define SLK_FSTAMP(i) do { SLK_STAMP_NR(i); if (a.stop > 0 && a.stop == (i)) return true; } while (0)
#define SLK_WSTAMP(w, i) do { if (tid == 64 * (w) && a.dbg) a.dbg[(size_t)bidx * 32 + (i)] = clock64(); } while (0)
#define SLK_FBAIL(code) do { SLK_NOTE(28, code); return false; } while (0)
#else
#define SLK_FSTAMP(i) do { } while (0)
#define SLK_WSTAMP(w, i) do { } while (0)
#define SLK_FBAIL(code) return false
#endif

namespace slk {

template <int K> struct FastShape {
    static constexpr int N = 12 + 6 * K, Nq = 13 + 7 * K, S = 2 * N + 1, NSO3 = K + 1;
    static constexpr int NT = (N + 15) / 16, NTL = NT * (NT + 1) / 2, NKS = (N + 3) / 4;
    static constexpr int NROT = 3 * NSO3;
    static constexpr int NP = 3 * K * K + 15 * K + 6;          // (block, column) pairs: sum over blocks of toff + 3
    static constexpr int NIT = NP + NSO3;                      // + the centre point of every block (a pair with l = 0)
    static constexpr int RND = (NIT + 255) / 256;
    // LDS carve, doubles
    static constexpr int oLt = 0;                              // tiled factor
    static constexpr int oMu = oLt + NTL * 256;                // mean (Nq)
    static constexpr int oRef = oMu + ((Nq + 7) & ~7);
    static constexpr int oDelta = oRef + ((Nq + 7) & ~7);      // 64
    static constexpr int oMd = oDelta + 64;                    // mean_delta in rotation-row space (32)
    static constexpr int oD0 = oMd + 32;                       // centre deviations (32)
    static constexpr int oD0s = oD0 + 32;                      // (S - 2 (toff + 3)) * centre deviation (32)
    static constexpr int oPd = oD0s + 32;                      // p [32], corrected centre deviations [32]
    static constexpr int oCq = oPd + 64;                       // ref_b^-1 mu_b (4 NSO3 -> 40)
    static constexpr int oStr = oCq + 40;                      // the two odd parts of row 15 that fall into tile (0, 1)
    static constexpr int oInts = oStr + 64;                    // 64 ints
    static constexpr int oTab = oInts + 32;                    // series coefficients (26)
    static constexpr int oU = oTab + 32;                       // union region
    // union, phases 1 - 5: Yp [64][17] (rows [y+ ; y-] of a column, one double of padding: the lanes of a wave write
    // different banks; later W [512] and the parked prefix sums), dZ interleaved [512], Sm, mdiag, b, innov, dz0, tmpS
    static constexpr int uYp = 0, uW = 0, uDZ = 1088, uSm = 1600, uMdiag = 1664, uB = 1728, uInnov = 1792, uDz0 = 1800, uTmp = 1808;
    static constexpr int USZ = (NKS * 128 > 1936) ? NKS * 128 : 1936;   // phases 6 - 7: E^ [NKS][2][64]
    static constexpr int total = oU + USZ;
};

static_assert(FastShape<8>::total * 8 * 4 <= 160 * 1024 && FastShape<7>::total * 8 * 4 <= 160 * 1024, "four workgroups per CU");

__host__ __device__ inline int fast_step_lds_doubles(int k)
{
    switch (k) {
    case 4: return FastShape<4>::total;
    case 5: return FastShape<5>::total;
    case 6: return FastShape<6>::total;
    case 7: return FastShape<7>::total;
    case 8: return FastShape<8>::total;
    default: return 0;
    }
}

__device__ __forceinline__ double wave_sum_f64(double x) { return readlane_f64(wave_inclusive_scan(x), 63); }

// rotation-row index (3 b + comp) of tangent row t, 31 (an all-zero row of E^ / p / d0) for vector rows and padding
template <int N> __device__ __forceinline__ int fast_rho(int t)
{
    if (t >= N) return 31;
    if (t < 12) return (t >= 3 && t < 6) ? t - 3 : 31;
    const int cc = (t - 12) / 6, r = (t - 12) - 6 * cc;
    return r >= 3 ? 3 * cc + r : 31;                 // 3 + 3 cc + (r - 3)
}
// storage index of a vector tangent row, -1 for rotation rows (State.hpp:141-149, :246-252, :384-396)
__device__ __forceinline__ int fast_vec_storage(int t)
{
    if (t < 12) return t < 3 ? t : (t < 6 ? -1 : t + 1);
    const int cc = (t - 12) / 6, r = (t - 12) - 6 * cc;
    return r < 3 ? 13 + 7 * cc + r : -1;
}

// ---- SO(3) exp / log with the series coefficients in an LDS table (slk_math.hpp has the same series with literal
// coefficients: inlined a dozen times they pin 26 registers for the whole kernel).  Several arguments are evaluated in
// lockstep so that every coefficient is fetched once.  A wave whose arguments are all small takes the series directly,
// otherwise (uniform branch) an angle-halving form that covers the whole domain the fast path can meet; exp beyond 4 rad
// hands the filter to the general body (flag ints[50], checked after the next barrier, before any global write).
//   T[0..5]   cos sqrt x            degree 6 in y = -x, x <= 1/4   (then 1)     } near-minimax coefficients of
//   T[6..10]  sin sqrt x / sqrt x   degree 5                       (then 1)     } tools/series_coefficients.py (relative
//   T[11..18] atan u / u            degree 8 in y = -u^2, u^2 <= 1/16 (then 1)  } errors 6e-18, 4e-17, 9e-18)
__device__ const double fast_series_table[26] = {
    0x1.1d8d32755f8fbp-29, 0x1.27e40964b47d4p-22, 0x1.a01a00fb1bc6fp-16, 0x1.6c16c16bdd04ep-10, 0x1.5555555555421p-5, 0x1.fffffffffffffp-2,
    0x1.ac53ce336f805p-26, 0x1.71dd113fb7905p-19, 0x1.a01a01061c190p-13, 0x1.11111110ecfb3p-7, 0x1.555555555548fp-3,
    0x1.78be0a9b1dd1fp-5, 0x1.0b3340fe2ed9ap-4, 0x1.3ab708d770276p-4, 0x1.7459b99bfc19bp-4, 0x1.c71c5f4b9c2adp-4, 0x1.24924907fa636p-3, 0x1.999999996d307p-3, 0x1.5555555555481p-2,
    0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

// q[i] = exp(v[i]) for NV rotation vectors.  Rotations below 1 rad (every lane of the wave): the series directly; otherwise
// the series for v / 4 and two quaternion squarings, exp(v) = (exp(v / 4)^2)^2 -- up to 4 rad; ok = false beyond.
template <int NV>
__device__ __forceinline__ bool so3_exp_tab(const double *T, const double (&v)[NV][3], Quat (&q)[NV])
{
    double y[NV], cc[NV], ss[NV];
    bool small = true, ok = true;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double x = 0.25 * (v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2]);
        small = small && (x < 0.25);
        ok = ok && (x < 4.0);
        y[i] = -x;
    }
    const bool wide = !__all(small);                 // (uniform over the wave)
    if (wide) {
#pragma unroll
        for (int i = 0; i < NV; ++i) y[i] *= 0.0625;   // |v / 4|^2 / 4
    }
    {
        const double c0 = T[0], s0 = T[6];
#pragma unroll
        for (int i = 0; i < NV; ++i) { cc[i] = c0; ss[i] = s0; }
    }
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        const double ck = T[k], sk = T[6 + k];
#pragma unroll
        for (int i = 0; i < NV; ++i) { cc[i] = fma(cc[i], y[i], ck); ss[i] = fma(ss[i], y[i], sk); }
    }
    {
        const double c5 = T[5];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            cc[i] = fma(cc[i], y[i], c5);
            cc[i] = fma(cc[i], y[i], 1.0);
            ss[i] = fma(ss[i], y[i], 1.0);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double m = 0.5 * ss[i], w = cc[i];           // exp(v) = (m v, w);  wide: exp(v / 4) = (m / 4 v, w)
        if (wide) {
            m *= 0.25;
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {      // (w, m v)^2 = (2 w^2 - 1, 2 w m v) for a unit quaternion
                const double w2 = fma(2.0 * w, w, -1.0);
                m = 2.0 * w * m;
                w = w2;
            }
        }
        q[i] = Quat{m * v[i][0], m * v[i][1], m * v[i][2], w};
    }
    return ok;
}
// d[i] = log(q[i]) in MTK's form 2 atan(|vec| / w) / |vec| * vec (q and -q give the same result).  Rotations below ~28
// degrees (every lane of the wave): the series of atan(u) / u directly, u = |vec| / w.  Otherwise by halved angles:
// with w >= 0 (else take -q), t1 = tan(theta / 4) = |vec| / (1 + w) <= 1, two more halvings t <- t / (1 + sqrt(1 + t^2))
// bring tan(theta / 16) below 0.2, theta = 16 atan(t3): every unit quaternion is inside this domain.
template <int NV>
__device__ __forceinline__ bool so3_log_tab(const double *T, const Quat (&q)[NV], double (&d)[NV][3])
{
    double y[NV], sc[NV], f[NV];
    bool small = true;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double n2 = q[i].x * q[i].x + q[i].y * q[i].y + q[i].z * q[i].z, w2 = q[i].w * q[i].w;
        small = small && (q[i].w > 0.0) && (n2 * 16.0 < w2);
        y[i] = n2;
    }
    const bool wide = !__all(small);                 // (uniform over the wave)
    if (!wide) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double r = __builtin_amdgcn_rcp(q[i].w);
            r = fma(fma(-q[i].w, r, 1.0), r, r);
            r = fma(fma(-q[i].w, r, 1.0), r, r);
            sc[i] = 2.0 * r;                         // log(q) = 2 / w * [atan(u) / u] * vec
            y[i] = -(y[i] * r * r);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double n2 = y[i], aw = fabs(q[i].w);
            double r = __builtin_amdgcn_rcp(1.0 + aw);
            r = fma(fma(-(1.0 + aw), r, 1.0), r, r);
            r = fma(fma(-(1.0 + aw), r, 1.0), r, r);
            double k = r, t2 = n2 * r * r;           // tan(theta / 4) = k |vec|, its square
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {      // t <- t / (1 + sqrt(1 + t^2))
                const double a1 = 1.0 + t2;
                double rs = __builtin_amdgcn_rsq(a1);
                rs = rs * fma(-0.5 * a1 * rs, rs, 1.5);
                rs = rs * fma(-0.5 * a1 * rs, rs, 1.5);
                const double den = fma(a1, rs, 1.0);  // 1 + sqrt(1 + t^2)
                double rd = __builtin_amdgcn_rcp(den);
                rd = fma(fma(-den, rd, 1.0), rd, rd);
                rd = fma(fma(-den, rd, 1.0), rd, rd);
                k *= rd;
                t2 *= rd * rd;
            }
            sc[i] = (q[i].w < 0.0) ? -16.0 * k : 16.0 * k;   // theta / |vec| = 16 atan(t3) / |vec| = 16 k [atan(t3) / t3]
            y[i] = -t2;
        }
    }
    {
        const double a0 = T[11];
#pragma unroll
        for (int i = 0; i < NV; ++i) f[i] = a0;
    }
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const double ak = T[11 + k];
#pragma unroll
        for (int i = 0; i < NV; ++i) f[i] = fma(f[i], y[i], ak);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f[i] = fma(f[i], y[i], 1.0);
        const double sv = sc[i] * f[i];
        d[i][0] = sv * q[i].x; d[i][1] = sv * q[i].y; d[i][2] = sv * q[i].z;
    }
    return true;
}

// exclusive prefix sums over the lanes (columns) of the entries E0 .. E0 + CNT - 1 of the packed lower triangle of a a^T
template <int E0, int CNT>
__device__ __forceinline__ void fast_scan_entries(const double (&av)[8], double (&G)[20])
{
#pragma unroll
    for (int e = 0; e < CNT; ++e) {
        constexpr int dummy = 0; (void)dummy;
        const int q = E0 + e;
        const int r = (q >= 1) + (q >= 3) + (q >= 6) + (q >= 10) + (q >= 15) + (q >= 21) + (q >= 28), cc = q - r * (r + 1) / 2;
        const double pr = av[r] * av[cc];
        G[e] = wave_inclusive_scan(pr) - pr;
    }
}

// (block, column) pairs of the mean loop, tabulated at compile time: one 64-bit word per pair
//   low dword : bits 0-12 / 13-25 offsets (doubles, from the start of LDS) of L'(toff, j) / L'(toff + 1, j) (an always-zero
//               element where the tile is not stored), 26-29 the SO(3) block, 30-31 1 / 2 = the odd part of component 0 goes
//               to str[15] / str[31] (row 15's columns 16 / 17, whose tile (0, 1) is not stored)
//   high dword: bits 0-12 offset of L'(toff + 2, j), 13-23 offset of component 0 in E^
template <int K> struct FastPairTable {
    unsigned long long v[FastShape<K>::NP];
    constexpr FastPairTable() : v{}
    {
        int p = 0;
        for (int b = 0; b <= K; ++b) {
            const int to = b ? 9 + 6 * b : 3;
            for (int j = 0; j < to + 3; ++j, ++p) {
                unsigned long long a[3] = {0, 0, 0};
                for (int cc = 0; cc < 3; ++cc) {
                    const int t = to + cc, I = t >> 4, Jc = j >> 4;
                    a[cc] = (unsigned long long)(FastShape<K>::oLt + (Jc <= I ? (I * (I + 1) / 2 + Jc) * 256 + (j & 15) * 16 + (t & 15) : 16));
                }
                const int rho = 3 * b;
                const unsigned long long e0 = (unsigned long long)(((j >> 2) * 2 + (rho >> 4)) * 64 + (j & 3) * 16 + (rho & 15));
                const unsigned long long sd = (b == 1 && j >= 16) ? (unsigned long long)(j - 15) : 0ull;
                v[p] = a[0] | (a[1] << 13) | ((unsigned long long)b << 26) | (sd << 30) | (a[2] << 32) | (e0 << 45);
            }
        }
    }
};
template <int K> struct FastPairTableHolder { static __device__ const FastPairTable<K> tab; };
template <int K> __device__ const FastPairTable<K> FastPairTableHolder<K>::tab = FastPairTable<K>();

// ldm_columns (slk_kernels.hpp) with the column's deviations a_j = 1/2 dZ_j fetched from LDS where they are used instead of
// held in sixteen registers beside the 36 prefix sums: lane j = column j, Gf = prefix sums, Sm = S (8 x 8), kept = rows
// that survived the gate.  Writes Wb[bw_idx(j, :)] = 1/2 w~_j and mdiag[j] = sqrt(d_j); false if Pk - K S K^T is not
// positive definite.
#define SLK_G(r, c) Gf[(r) * ((r) + 1) / 2 + (c)]
__device__ __forceinline__ bool fast_ldm_columns(double (&Gf)[36], const double *dZi, const double *Sm, unsigned kept, int lane, int N,
                                                 double *Wb, double *mdiag)
{
    const bool live = lane < N;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool kr = (kept >> r) & 1u;
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            const bool in = kr && ((kept >> c) & 1u);
            const double sv = Sm[in ? r + 8 * c : 0];
            SLK_G(r, c) = in ? sv - SLK_G(r, c) : ((r == c) ? 1.0 : 0.0);
        }
    }
    // T_j = R R^T in place (R lower, its diagonal kept as reciprocals), y = R^-1 b, d = 1 - |y|^2, w = -R^-T y / sqrt(d)
    double y[8];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double d = SLK_G(j, j);
#pragma unroll
        for (int p = 0; p < j; ++p) d = fma(-SLK_G(j, p), SLK_G(j, p), d);
        ok = ok && (d > 0.0);
        double sq, rs;
        rsqrt_pivot(d, sq, rs);
        SLK_G(j, j) = rs;
#pragma unroll
        for (int i = j + 1; i < 8; ++i) {
            double v = SLK_G(i, j);
#pragma unroll
            for (int p = 0; p < j; ++p) v = fma(-SLK_G(i, p), SLK_G(j, p), v);
            SLK_G(i, j) = v * rs;
        }
        const double aj = dZi[(j >> 1) * 128 + 2 * lane + (j & 1)];
        double sacc = ((kept >> j) & 1u) ? 0.5 * aj : 0.0;
#pragma unroll
        for (int p = 0; p < j; ++p) sacc = fma(-SLK_G(j, p), y[p], sacc);
        y[j] = sacc * rs;
    }
    double dj = 1.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) dj = fma(-y[c], y[c], dj);
    ok = ok && (dj > 0.0);
    double sqd, rsd;
    rsqrt_pivot(dj, sqd, rsd);
    const double sc = -0.5 * rsd;
#pragma unroll
    for (int c = 7; c >= 0; --c) {                   // w overwrites y from the back
        double sacc = y[c];
#pragma unroll
        for (int p = c + 1; p < 8; ++p) sacc = fma(-SLK_G(p, c), y[p], sacc);
        y[c] = sacc * SLK_G(c, c);
        Wb[bw_idx(lane, c)] = live ? y[c] * sc : 0.0;
    }
    mdiag[lane] = live ? sqd : 1.0;
    return __all(ok) != 0;
}
#undef SLK_G

// L <- L M on the matrix cores, tiled factor (see ldm_product in slk_kernels.hpp for the algebra).  JB = tile column of the
// last column any measurement row depends on: beyond it M is the identity (dZ rows are exactly zero), those blocks are
// skipped.  fill() runs between the two barriers (everything but the factor is dead there).
template <int NT, class FillFn>
__device__ __forceinline__ void ldm_product_tiled(double *Lt, const double *dZi, const double *Wb, const double *mdiag, int lane,
                                                  int wave, int JB, FillFn fill)
{
    const int c = lane & 15, g = lane >> 4;
    const int lw = (g >> 1) * 128 + 2 * c + (g & 1);           // lane part of bw_idx(16 X + c, g)
    constexpr int MAXT = (NT == 4) ? 4 : 2;
    d4 acc[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            if (ldm_tile_wave<NT>(I, J) != wave || J > JB) continue;
            d4 a = {0.0, 0.0, 0.0, 0.0};
            const double wf0 = Wb[32 * J + lw], wf1 = Wb[256 + 32 * J + lw];
#pragma unroll
            for (int Kb = J; Kb <= I; ++Kb) {
                if (Kb > JB) continue;
                d4 mt = {0.0, 0.0, 0.0, 0.0};
                const double a0 = dZi[32 * Kb + lw], a1 = dZi[256 + 32 * Kb + lw];
                double lf[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) lf[s] = Lt[(I * (I + 1) / 2 + Kb) * 256 + 64 * s + lane];
                mt = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, wf0, mt, 0, 0, 0);
                mt = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, wf1, mt, 0, 0, 0);
                if (Kb == J) {
                    const double dg = mdiag[16 * J + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = g + 4 * r;
                        mt[r] = (row > c) ? mt[r] : ((row == c) ? dg : 0.0);
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) a = __builtin_amdgcn_mfma_f64_16x16x4f64(mt[s], lf[s], a, 0, 0, 0);
            }
            acc[ldm_tile_slot<NT>(I, J)] = a;
        }
    __syncthreads();                                  // every wave has read what it needs of the old factor
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            if (ldm_tile_wave<NT>(I, J) != wave || J > JB) continue;
            const d4 a = acc[ldm_tile_slot<NT>(I, J)];
#pragma unroll
            for (int r = 0; r < 4; ++r) Lt[(I * (I + 1) / 2 + J) * 256 + 64 * r + lane] = a[r];
        }
    fill();
    __syncthreads();
}

// One tile column J of P+ = O O^T + E^ E^^T (k-steps 0 .. 4 J + 3; tile column 0 also the step that holds row 15's columns
// 16 / 17): acc[I - J] += frag(I) frag(J)^T.
template <int K, int J>
__device__ __forceinline__ void fast_rebuild_col(const double *Lt, const double *Et, const double *str, int lane,
                                                 d4 (&acc)[FastShape<K>::NT])
{
    using F = FastShape<K>;
    constexpr int NT = F::NT, N = F::N;
    constexpr int KEND = (4 * J + 4 + (J == 0 ? 1 : 0) < F::NKS) ? 4 * J + 4 + (J == 0 ? 1 : 0) : F::NKS;
    const int c = lane & 15, g = lane >> 4;
    int eoff[NT];
#pragma unroll
    for (int I = J; I < NT; ++I) {
        const int rho = fast_rho<N>(16 * I + c);
        eoff[I] = (rho >> 4) * 64 + g * 16 + (rho & 15);
    }
#pragma unroll
    for (int ks = 0; ks < KEND; ++ks) {
        double fo[NT], fe[NT];
#pragma unroll
        for (int I = J; I < NT; ++I) {
            const int Kc = ks >> 2;
            fo[I] = (Kc <= I) ? Lt[(I * (I + 1) / 2 + Kc) * 256 + (ks & 3) * 64 + lane] : str[lane];
            fe[I] = Et[ks * 128 + eoff[I]];
        }
#pragma unroll
        for (int I = J; I < NT; ++I) {
            acc[I - J] = __builtin_amdgcn_mfma_f64_16x16x4f64(fo[I], fo[J], acc[I - J], 0, 0, 0);
            acc[I - J] = __builtin_amdgcn_mfma_f64_16x16x4f64(fe[I], fe[J], acc[I - J], 0, 0, 0);
        }
    }
}

// ... + p d0^T + d0 p^T (one k-step: A = [p, d0], B = [d0, p]) and out: the mirror triangle straight from the accumulators,
// the lower triangle of the off-diagonal tiles through a private 16 x 17 transpose buffer.
template <int K, int J>
__device__ __forceinline__ void fast_store_col(const double *pd, double *bufs, double *oP, int lane, d4 (&acc)[FastShape<K>::NT], bool lower_only)
{
    using F = FastShape<K>;
    constexpr int NT = F::NT, N = F::N;
    const int c = lane & 15, g = lane >> 4;
    {
        const int rj = fast_rho<N>(16 * J + c);
        const double bf = pd[g == 0 ? 32 + rj : (g == 1 ? rj : 31)];
#pragma unroll
        for (int I = J; I < NT; ++I) {
            const int ri = fast_rho<N>(16 * I + c);
            const double af = pd[g == 0 ? ri : (g == 1 ? 32 + ri : 31)];
            acc[I - J] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[I - J], 0, 0, 0);
        }
    }
    double *o = oP + c + g * N;                      // lane part of both orientations
#pragma unroll
    for (int I = J; I < NT; ++I) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {                // acc[r] of lane (c, g) = P+(16 I + g + 4 r, 16 J + c) -> P(col, row)
            const bool full = 16 * I + 4 * r + 3 < N && 16 * J + 15 < N;
            if (I > J && lower_only) continue;       // (the mirror triangle is brought up to date when somebody wants it)
            if (16 * I + 4 * r < N && (full || (16 * I + 4 * r + g < N && 16 * J + c < N)))
                o[16 * J + (16 * I + 4 * r) * N] = acc[I - J][r];
        }
        if (I > J) {
            double *buf = bufs + (I * (I - 1) / 2 + J) * 272;
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[c * 17 + g + 4 * r] = acc[I - J][r];     // buf[col][row]
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // lane (c, g): row 16 I + c, column 16 J + g + 4 q
                const double v = buf[(g + 4 * q) * 17 + c];
                const bool full = 16 * I + 15 < N;   // (columns of an off-diagonal tile are always inside)
                if (full || 16 * I + c < N) o[16 * I + (16 * J + 4 * q) * N] = v;
            }
        }
    }
}

template <int K>
__device__ __forceinline__ bool msckf_step_fast(const KArgs &a, double *smem)
{
    using F = FastShape<K>;
    constexpr int N = F::N, Nq = F::Nq, S = F::S, NSO3 = F::NSO3, NT = F::NT, NKS = F::NKS, NROT = F::NROT;
    constexpr int NP = F::NP;
    if (a.mm != SLK_MM_FEATURE_PROJ || a.gate == 2 || a.emit != 0 || a.rebuild_prec != 0 || !a.mp || a.m != 8) return false;
    const int bidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    double *Lt = smem + F::oLt, *mu = smem + F::oMu, *ref = smem + F::oRef, *delta = smem + F::oDelta, *md32 = smem + F::oMd;
    double *d0 = smem + F::oD0, *d0s = smem + F::oD0s, *pd = smem + F::oPd, *cq = smem + F::oCq, *str = smem + F::oStr, *U = smem + F::oU;
    int *ints = reinterpret_cast<int *>(smem + F::oInts);          // [0..39] gate lists of the waves, [48] [49] flags
    double *T = smem + F::oTab;
    double *Yp = U + F::uYp, *Wb = U + F::uW, *dZi = U + F::uDZ, *Sm = U + F::uSm, *mdiag = U + F::uMdiag, *bvec = U + F::uB;
    double *innov = U + F::uInnov, *dz0 = U + F::uDz0, *tmpS = U + F::uTmp, *Et = U;
    const double *gmean = a.mean + (size_t)bidx * Nq;
    const double *gP = a.P + (size_t)bidx * N * N;
    const double *mp = a.mp + (size_t)bidx * a.mp_stride;
    SLK_STAMP_NR(0);

    // ---- phase 0: mean, tiled factor, small arrays.  Every global load is issued before the first use; the conditions that
    // hand the filter to the general body -- failed first factorisation, pose index out of range (SLK_ST_BAD_INDEX there),
    // a rotation column that may exceed pi (Msckf.hpp:407-413: covXZ = L A would not hold) -- meet in one barrier
    int jmax = 0, bad;
    // the factor: from msckf_chol_kernel's workspace, or (a.wsfail == nullptr) factored HERE by wave 0 -- cholp_factor, panel
    // by rows, straight into the tiles -- while the other waves fetch the small arrays
#ifndef SLK_MSCKF_FACTOR_KERNEL     // (-DSLK_MSCKF_FACTOR_KERNEL: the three-launch form, the factor through the workspace only -- same step time,
    const bool here = a.wsfail == nullptr;       //  1.28 x instead of 0.74 x the algorithmic bytes: profiles/r03_ab_msckf_factor_inside.log)
#else
    constexpr bool here = false;
    if (!a.wsfail) return false;
#endif
    {
        const double *gL = here ? gP : a.wsL + (size_t)bidx * pk_size(N);
        const int t = lane, It = t >> 4;
        const int lbase = (It * (It + 1) / 2) * 256 + wave * 16 + (t & 15);
        double v[16];
        if (!here) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = 4 * u + wave;
                const bool in = j < N && t >= j && t < N;
                v[u] = gL[in ? pkcol(N, j) + t : 0];
            }
        }
        const double mu0 = (tid < Nq) ? gmean[tid] : 0.0;
        const int t0 = (tid && tid < NSO3) ? 9 + 6 * tid : 3;
        const double pdg = gP[t0 * (N + 1)] + gP[(t0 + 1) * (N + 1)] + gP[(t0 + 2) * (N + 1)];
        const double tabv = fast_series_table[tid < 26 ? tid : 25];
        bad = here ? 0 : (a.wsfail[bidx] >= 0);
        bool ok = true;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const double cf = mp[4 * f + 3];
            ok = ok && (cf >= 0.0) && (cf <= (double)K);
            const int cp = ok ? (int)cf : 0, tp = cp ? 6 + 6 * cp : 0;
            jmax = (tp + 5 > jmax) ? tp + 5 : jmax;
        }
        bad |= !ok;
        if (tid < NSO3) bad |= !(pdg < 9.869604401089358);
        if (tid < Nq) mu[tid] = mu0;
        if (tid < 64) { str[tid] = 0.0; pd[tid] = 0.0; ints[tid] = 0; }
        if (tid < 32) { d0[tid] = 0.0; md32[tid] = 0.0; }
        if (tid < 26) T[tid] = tabv;
        if (!here) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = 4 * u + wave;
                const bool in = j < N && t >= j && t < N;
                if ((u >> 2) < NT && It >= (u >> 2) && It < NT) Lt[lbase + (u >> 2) * 256 + (u & 3) * 64] = in ? v[u] : 0.0;
            }
        }
        // (after the stores above: the sixteen loaded values must not live across this region)
        if (here && wave == 0) {
            d4 fa[CholM<NT>::NTL];
            cholm_load_t<NT>(fa, N, lane, [&](int i, int j) { return gP[i + (size_t)j * N]; });
            bad |= cholp_factor<NT, 2>(fa, Lt, N, U, lane) >= 0;      // (the union region is free until phase 1: its colbuf)
        }
    }
    if (__syncthreads_or(bad)) SLK_FBAIL(2);
    SLK_FSTAMP(3);

    // ---- phase 1: Z = h(X) (Msckf.hpp:231-232), one wave per feature, lane j = the sigma pair of column j
    {
        const int f = wave;
        const int cp = (int)mp[4 * f + 3];
        const int tp = cp ? 6 + 6 * cp : 0, sp = cp ? 6 + 7 * cp : 0;
        const double fx = mp[4 * f], fy = mp[4 * f + 1], fz = mp[4 * f + 2];
        const int j = lane, Jc = j >> 4;
        const bool act = j < tp + 6;                            // column j reaches the pose's six rows
        double l[6], x[7];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int t = tp + q, I = t >> 4;
            const bool in = act && Jc <= I;
            l[q] = Lt[in ? (I * (I + 1) / 2 + Jc) * 256 + (j & 15) * 16 + (t & 15) : 16];     // Lt[16] = L(0, 1) = 0
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) x[q] = mu[sp + q];
        double zp0, zp1, zm0, zm1;
        {
            const double rv[2][3] = {{l[3], l[4], l[5]}, {-l[3], -l[4], -l[5]}};
            Quat ex[2];
            if (!__all(so3_exp_tab<2>(T, rv, ex))) ints[50] = 1; // a rotation column beyond the series' domain (1 rad)
            const Quat qx = Quat{x[3], x[4], x[5], x[6]};
            double lx, ly, lz;
            // (x / z by one refined reciprocal and a correction step: within an ulp of the IEEE quotient, half the instructions
            // of two divisions)
            auto quot = [](double lx_, double ly_, double lz_, double &q0, double &q1) __attribute__((always_inline)) {
                double r = __builtin_amdgcn_rcp(lz_);
                r = fma(fma(-lz_, r, 1.0), r, r);
                r = fma(fma(-lz_, r, 1.0), r, r);
                q0 = lx_ * r; q0 = fma(fma(-lz_, q0, lx_), r, q0);
                q1 = ly_ * r; q1 = fma(fma(-lz_, q1, ly_), r, q1);
            };
            qrot(qconj(qmul(qx, ex[0])), fx - (x[0] + l[0]), fy - (x[1] + l[1]), fz - (x[2] + l[2]), lx, ly, lz);
            quot(lx, ly, lz, zp0, zp1);
            qrot(qconj(qmul(qx, ex[1])), fx - (x[0] - l[0]), fy - (x[1] - l[1]), fz - (x[2] - l[2]), lx, ly, lz);
            quot(lx, ly, lz, zm0, zm1);
        }
        const double Z00 = readlane_f64(zp0, 63), Z01 = readlane_f64(zp1, 63);       // lanes >= tp + 6 evaluate X_0
        const double yp0 = zp0 - Z00, yp1 = zp1 - Z01, ym0 = zm0 - Z00, ym1 = zm1 - Z01;
        Yp[j * 17 + 2 * f] = yp0; Yp[j * 17 + 2 * f + 1] = yp1;
        Yp[j * 17 + 8 + 2 * f] = ym0; Yp[j * 17 + 8 + 2 * f + 1] = ym1;
        dZi[f * 128 + 2 * j] = yp0 - ym0; dZi[f * 128 + 2 * j + 1] = yp1 - ym1;
        const double s0 = wave_sum_f64(yp0 + ym0), s1 = wave_sum_f64(yp1 + ym1);
        if (lane == 0) {                                        // mean_z (:234) about Z_0, innovation (:236)
            const double e0 = s0 / (double)S, e1 = s1 / (double)S;
            innov[2 * f] = a.z[(size_t)bidx * 8 + 2 * f] - (Z00 + e0);
            innov[2 * f + 1] = a.z[(size_t)bidx * 8 + 2 * f + 1] - (Z01 + e1);
            dz0[2 * f] = e0; dz0[2 * f + 1] = e1;
        }
    }
    __syncthreads();
    if (ints[50]) SLK_FBAIL(3);
    SLK_FSTAMP(4);

    // ---- phase 2: S = 1/2 sum (Z_i - mean_z)(Z_i - mean_z)^T + R (:238) on wave 1
    double Gs[20];
    if (wave == 1) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const double fr = Yp[ks * 68 + g4 * 17 + c16];      // rows [y+ ; y-] of column 4 ks + g
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fr, fr, acc, 0, 0, 0);
        }
        if (c16 < 8) { tmpS[g4 + 8 * c16] = acc[0]; tmpS[g4 + 4 + 8 * c16] = acc[1]; }
        else { tmpS[64 + g4 + 8 * (c16 - 8)] = acc[2]; tmpS[64 + g4 + 4 + 8 * (c16 - 8)] = acc[3]; }
        wave_sync();
        const double *R = a.R + (size_t)bidx * a.r_stride;
        const int row = lane & 7, col = lane >> 3;
        Sm[lane] = 0.5 * ((tmpS[lane] + tmpS[64 + lane]) - (double)S * dz0[row] * dz0[col]) + R[lane];
    } else if (wave >= 2) {
        // prefix sums over the columns of a a^T (ldm_prefix: the 36 entries of the packed lower triangle, lane = column),
        // the first 16 by wave 2, the others by wave 3: one wave's 36 scans were the longest serial stretch of the step
        double av[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) av[cc] = 0.5 * dZi[(cc >> 1) * 128 + 2 * lane + (cc & 1)];
        if (wave == 2) fast_scan_entries<0, 16>(av, Gs);
        else fast_scan_entries<16, 20>(av, Gs);
        SLK_WSTAMP(3, 21);
    }
    __syncthreads();
    SLK_FSTAMP(6);

    // ---- gate: removeOutliers (:723-754) with the shifted second erase (:741-744); every wave takes the same decisions
    unsigned kept = 0xffu, nout = 0u;
    if (a.gate) {
        const int p = 2 * (lane & 3), q = p + 1;
        const double s00 = Sm[p + 8 * p], s01 = Sm[p + 8 * q], s10 = Sm[q + 8 * p], s11 = Sm[q + 8 * q];
        const double det = s00 * s11 - s01 * s10, r0 = innov[p], r1 = innov[q];
        const double d2 = (r0 * (s11 * r0 - s01 * r1) + r1 * (s00 * r1 - s10 * r0)) / det;
        if (!__all(d2 < 5.99)) {                                // chi2_0.95(2), Msckf.hpp:861-865
            int *ix = ints + 10 * wave;
            if (lane == 0) {
                int cnt = 8, i = 0;
                unsigned no = 0;
                for (int r = 0; r < 8; ++r) ix[r] = r;
                while (i < cnt / 2) {
                    const int pp = ix[2 * i], qq = ix[2 * i + 1];
                    const double t00 = Sm[pp + 8 * pp], t01 = Sm[pp + 8 * qq], t10 = Sm[qq + 8 * pp], t11 = Sm[qq + 8 * qq];
                    const double dt = t00 * t11 - t01 * t10, u0 = innov[pp], u1 = innov[qq];
                    const double dd = (u0 * (t11 * u0 - t01 * u1) + u1 * (t00 * u1 - t10 * u0)) / dt;
                    if (!(dd < 5.99)) {
                        for (int rep = 0; rep < 2; ++rep) {     // removeRow semantics, :688-697
                            const int pos = 2 * i + rep, numRows = cnt - 1;
                            if (pos < numRows) for (int w = pos; w < numRows; ++w) ix[w] = ix[w + 1];
                            cnt = numRows;
                        }
                        no++;
                    } else {
                        i++;
                    }
                }
                unsigned kp = 0;
                for (int r = 0; r < cnt; ++r) kp |= 1u << ix[r];
                ix[8] = (int)kp; ix[9] = (int)no;
            }
            wave_sync();
            kept = (unsigned)ix[8]; nout = (unsigned)ix[9];
        }
    }
    if (kept == 0u) SLK_FBAIL(4);                               // every block rejected (:250): status by the general body
    SLK_FSTAMP(7);

    // ---- gain side.  wave 3: columns of the factor update.  wave 0: x = S^-1 nu over the surviving rows (rejected rows
    // as identity rows), b = 1/2 dZ x, delta = K nu = L b (:257, :263), then the reference point of the mean loop.
    if (wave == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Yp[e * 64 + lane] = Gs[e];        // (the Yp rows are dead: S is done)
    }
    __syncthreads();
    if (wave == 0) {
        double gg[8][8], gi[8], y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const bool in = ((kept >> i) & 1u) && ((kept >> j) & 1u);
                const double v = Sm[in ? i + 8 * j : 0];
                gg[i][j] = in ? v : (i == j ? 1.0 : 0.0);
            }
        bool spd = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double d = gg[j][j];
#pragma unroll
            for (int p = 0; p < j; ++p) d = fma(-gg[j][p], gg[j][p], d);
            spd = spd && (d > 0.0);
            double sq, rs;
            rsqrt_pivot(d, sq, rs);
            gi[j] = rs;
#pragma unroll
            for (int i = j + 1; i < 8; ++i) {
                double v = gg[i][j];
#pragma unroll
                for (int p = 0; p < j; ++p) v = fma(-gg[i][p], gg[j][p], v);
                gg[i][j] = v * rs;
            }
            double s = ((kept >> j) & 1u) ? innov[j] : 0.0;
#pragma unroll
            for (int p = 0; p < j; ++p) s = fma(-gg[j][p], y[p], s);
            y[j] = s * rs;
        }
#pragma unroll
        for (int cc = 7; cc >= 0; --cc) {
            double s = y[cc];
#pragma unroll
            for (int p = cc + 1; p < 8; ++p) s = fma(-gg[p][cc], y[p], s);
            y[cc] = s * gi[cc];
        }
        if (!spd && lane == 0) ints[49] = 1;
        SLK_WSTAMP(0, 24);
        double bj = 0.0;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) bj = fma(dZi[(cc >> 1) * 128 + 2 * lane + (cc & 1)], y[cc], bj);
        bvec[lane] = 0.5 * bj;                                   // rows >= N of dZ are zero
        wave_sync();
        {
            const int t = lane, It = t >> 4;
            const double *Lrow = Lt + (It * (It + 1) / 2) * 256 + (t & 15);
            double part[NT];
#pragma unroll
            for (int Jb = 0; Jb < NT; ++Jb) {
                double s = 0.0;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    if (16 * Jb + jj < N) s = fma(Lrow[Jb * 256 + jj * 16], bvec[16 * Jb + jj], s);
                    if ((jj & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // (eight columns of loads in flight at a time: registers)
                }
                part[Jb] = s;
            }
            double dl = part[0];
#pragma unroll
            for (int Jb = 1; Jb < NT; ++Jb) dl += (It >= Jb) ? part[Jb] : 0.0;      // (tile columns right of the row's own hold other tiles)
            if (t < 64) delta[t] = (t < N) ? dl : 0.0;
            const int s = (t < N) ? fast_vec_storage(t) : -1;
            if (s >= 0) ref[s] = mu[s] + dl;                     // X_0 = mu [+] delta (:501), vector part
        }
        wave_sync();
        if (lane < NSO3) {
            const int to = lane ? 9 + 6 * lane : 3, so = lane ? 9 + 7 * lane : 3;
            const Quat qm = ldq(mu + so);
            const double dv[1][3] = {{delta[to], delta[to + 1], delta[to + 2]}};
            Quat ex[1];
            if (!so3_exp_tab<1>(T, dv, ex)) ints[50] = 1;
            const Quat qr = qmul(qm, ex[0]);
            stq(ref + so, qr);
            stq(cq + 4 * lane, qmul(qconj(qr), qm));
        }
        SLK_WSTAMP(0, 25);
    }
    else if (wave == 3) {
        SLK_WSTAMP(3, 22);
        double Gf[36];
#pragma unroll
        for (int e = 0; e < 16; ++e) Gf[e] = Yp[e * 64 + lane];
#pragma unroll
        for (int e = 16; e < 36; ++e) Gf[e] = Gs[e - 16];
        const bool pdok = fast_ldm_columns(Gf, dZi, Sm, kept, lane, N, Wb, mdiag);
        if (!pdok && lane == 0) ints[48] = 1;
        SLK_WSTAMP(3, 23);
    }
    __syncthreads();
    if (ints[48] | ints[49] | ints[50]) SLK_FBAIL(5);                      // indefinite downdate / non-SPD S / large rotation: the general body decides
#ifdef SLK_EXP_A
    return true;
#endif
    SLK_FSTAMP(8);

    // ---- applyDelta's factor: L' = L chol(I - B B^T) (:262-263, :659-662)
    unsigned long long pt0 = 0, pt1 = 0;                       // pair descriptors of the mean loop
    ldm_product_tiled<NT>(Lt, dZi, Wb, mdiag, lane, wave, jmax >> 4, [&]() {
        pt0 = FastPairTableHolder<K>::tab.v[tid < NP ? tid : 0];
        pt1 = FastPairTableHolder<K>::tab.v[(wave == 1 && 256 + lane < NP) ? 256 + lane : 0];
        for (int e = tid; e < NKS * 128; e += 256) Et[e] = 0.0;
    });
    SLK_FSTAMP(11);

    // ---- manifold mean of the re-drawn sigma points (:664 -> :499-525) over (block, column) pairs: round 0 on every wave,
    // the pairs beyond 256 on wave 1, the centre points (one deviation each) on wave 2, the row sums on waves 2 / 3
    constexpr int RP = (NP + 255) / 256;                         // rounds of pairs (2 for k >= 7)
    int aL[RP][3], aE0[RP], bsd[RP];                            // (component 1 / 2 of E^ and the row-15 case are derived at use)
    bool val[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const unsigned long long w = r ? pt1 : pt0;
        const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
        val[r] = (r ? 256 + lane : tid) < NP && (r == 0 || wave == 1);
        bsd[r] = lo >> 26;                                       // block | odd-part-to-str case << 4
        aL[r][0] = lo & 0x1fff; aL[r][1] = (lo >> 13) & 0x1fff; aL[r][2] = hi & 0x1fff;
        aE0[r] = (hi >> 13) & 0x7ff;
    }
    auto e_off = [](int e0, int b, int cc) __attribute__((always_inline)) {     // offset of component cc in E^
        const int rho = 3 * b;                                   // (a component that crosses into the second 16 rows: + 49)
        const int e1 = e0 + ((rho & 15) == 15 ? 49 : 1);
        return cc == 0 ? e0 : (cc == 1 ? e1 : e1 + (((rho + 1) & 15) == 15 ? 49 : 1));
    };
    double dpl[RP][3], dmi[RP][3], dc[3] = {0.0, 0.0, 0.0};
    int it = 0;
    for (;;) {
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            if (r == 0 ? 64 * wave >= NP : wave != 1) continue;  // nothing for this wave in this round
            const int b = bsd[r] & 15, to = b ? 9 + 6 * b : 3;
            const double l0 = smem[aL[r][0]], l1 = smem[aL[r][1]], l2 = smem[aL[r][2]];
            const double e0 = delta[to], e1 = delta[to + 1], e2 = delta[to + 2];
            const Quat cb = ldq(cq + 4 * b);
            const double rv[2][3] = {{e0 + l0, e1 + l1, e2 + l2}, {e0 - l0, e1 - l1, e2 - l2}};
            Quat ex[2];
            double dd[2][3];
            bool ok = so3_exp_tab<2>(T, rv, ex);
            ex[0] = qmul(cb, ex[0]);
            ex[1] = qmul(cb, ex[1]);
#ifdef SLK_STAMPS
            {   // diagnostic: which waves leave the direct series (bit 0 / 1 of round r: exp / log went the angle-halving way)
                bool se = true, sl = true;
                for (int i = 0; i < 2; ++i) {
                    se = se && (0.25 * (rv[i][0] * rv[i][0] + rv[i][1] * rv[i][1] + rv[i][2] * rv[i][2]) < 0.25);
                    const double n2 = ex[i].x * ex[i].x + ex[i].y * ex[i].y + ex[i].z * ex[i].z;
                    sl = sl && (ex[i].w > 0.0) && (n2 * 16.0 < ex[i].w * ex[i].w);
                }
                const int code = (__all(se) ? 0 : 1) | (__all(sl) ? 0 : 2);
                const int slot = wave < 2 ? 26 + wave : 27 + wave;
                if (lane == 0 && a.dbg && it == 0) a.dbg[(size_t)bidx * 32 + slot] = (r ? a.dbg[(size_t)bidx * 32 + slot] : 0) | (code << (2 * r));
            }
#endif
            ok = so3_log_tab<2>(T, ex, dd) && ok;
            if (!__all(ok)) ints[50] = 1;                        // beyond the series' domains
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) { dpl[r][cc] = dd[0][cc]; dmi[r][cc] = dd[1][cc]; }
            if (val[r]) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) Et[e_off(aE0[r], b, cc)] = 0.5 * (dpl[r][cc] + dmi[r][cc]);
            }
        }
        if (wave == 2) {                                         // X_0 [-] ref per block
            const int b = lane < NSO3 ? lane : 0, to = b ? 9 + 6 * b : 3;
            const double rv[1][3] = {{delta[to], delta[to + 1], delta[to + 2]}};
            const Quat cb = ldq(cq + 4 * b);
            Quat ex[1];
            double dd[1][3];
            bool ok = so3_exp_tab<1>(T, rv, ex);
            ex[0] = qmul(cb, ex[0]);
            ok = so3_log_tab<1>(T, ex, dd) && ok;
            if (!__all(ok)) ints[50] = 1;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                dc[cc] = dd[0][cc];
                if (lane < NSO3) {                               // (the points beyond the block's columns: weight S - 2 (toff + 3))
                    d0[3 * b + cc] = dc[cc];
                    d0s[3 * b + cc] = (double)(S - 2 * (b ? 12 + 6 * b : 6)) * dc[cc];
                }
            }
        }
        __syncthreads();
        if (ints[50]) SLK_FBAIL(6);                              // (nothing has been written yet: the general body starts over)
        // mean_delta = sum_i (X_i [-] ref) / S (:507-509): the S - 2 (toff + 3) points beyond the block's columns equal X_0
        if (wave >= 2) {
            const int hr = wave - 2;
            double s = 0.0;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) s += Et[(ks * 2 + hr) * 64 + lane];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const int rho = 16 * hr + lane;                      // (lanes 0 .. 15 hold the sums of the rows 16 hr ..)
            if (lane < 16 && rho < NROT) md32[rho] = (2.0 * s + d0s[rho]) * (1.0 / (double)S);
        }
        __syncthreads();
        const double mdl = (lane < NROT) ? md32[lane] : 0.0;
        const double norm = sqrt(wave_sum_f64(mdl * mdl));
        if (wave == 0 && lane < NSO3) {                          // reference += mean_delta (:510)
            const int so = lane ? 9 + 7 * lane : 3;
            const double dv[1][3] = {{md32[3 * lane], md32[3 * lane + 1], md32[3 * lane + 2]}};
            Quat ex[1];
            if (!so3_exp_tab<1>(T, dv, ex)) ints[50] = 1;        // (a mean_delta beyond 1 rad)
            const Quat qr = qmul(ldq(ref + so), ex[0]);
            stq(ref + so, qr);
            stq(cq + 4 * lane, qmul(qconj(qr), ldq(mu + so)));
        }
        if (!(norm > 1e-6)) break;                               // :511
        if (++it >= 64) SLK_FBAIL(7);                            // (nothing has been written yet: the general body starts over)
        __syncthreads();
    }
    SLK_FSTAMP(12);
    SLK_NOTE(20, it + 1);
    // The loop leaves with |mean_delta| <= 1e-6: deviations against the FINAL mean by the first-order correction
    //     d' = d - Jl^-1(d) m,  Jl^-1(d) m = m - 1/2 d x m + (1/12 + |d|^2 / 720) d x (d x m)       (slk_kernels.hpp)
    // then the odd parts into the rotation rows of the factor array, the even parts minus the centre into E^.
    {
        auto fix = [&](double &x, double &y, double &z, double m0, double m1, double m2) __attribute__((always_inline)) {
            const double cx = y * m2 - z * m1, cy = z * m0 - x * m2, cz = x * m1 - y * m0;      // d x m
            const double ax = y * cz - z * cy, ay = z * cx - x * cz, az = x * cy - y * cx;      // d x (d x m)
            const double a12 = 1.0 / 12.0 + (x * x + y * y + z * z) * (1.0 / 720.0);
            x = x - m0 + 0.5 * cx - a12 * ax;
            y = y - m1 + 0.5 * cy - a12 * ay;
            z = z - m2 + 0.5 * cz - a12 * az;
        };
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            if (r == 0 ? 64 * wave >= NP : wave != 1) continue;
            const int b = bsd[r] & 15, sd = bsd[r] >> 4;
            const double m0 = md32[3 * b], m1 = md32[3 * b + 1], m2 = md32[3 * b + 2];
            double c0 = d0[3 * b], c1 = d0[3 * b + 1], c2 = d0[3 * b + 2];
            fix(c0, c1, c2, m0, m1, m2);
            fix(dpl[r][0], dpl[r][1], dpl[r][2], m0, m1, m2);
            fix(dmi[r][0], dmi[r][1], dmi[r][2], m0, m1, m2);
            if (val[r]) {
                smem[sd ? F::oStr + 16 * sd - 1 : aL[r][0]] = 0.5 * (dpl[r][0] - dmi[r][0]);    // (row 15's columns 16 / 17: str)
                smem[aL[r][1]] = 0.5 * (dpl[r][1] - dmi[r][1]);
                smem[aL[r][2]] = 0.5 * (dpl[r][2] - dmi[r][2]);
                Et[e_off(aE0[r], b, 0)] = 0.5 * (dpl[r][0] + dmi[r][0]) - c0;
                Et[e_off(aE0[r], b, 1)] = 0.5 * (dpl[r][1] + dmi[r][1]) - c1;
                Et[e_off(aE0[r], b, 2)] = 0.5 * (dpl[r][2] + dmi[r][2]) - c2;
            }
        }
        if (wave == 2 && lane < NSO3) {
            fix(dc[0], dc[1], dc[2], md32[3 * lane], md32[3 * lane + 1], md32[3 * lane + 2]);
            pd[32 + 3 * lane] = dc[0]; pd[32 + 3 * lane + 1] = dc[1]; pd[32 + 3 * lane + 2] = dc[2];
        }
    }
    __syncthreads();
    if (ints[50]) SLK_FBAIL(8);                                  // (the last move of the reference was beyond 1 rad)
    // the new mean (:664): no fallback beyond this point
    double *omean = a.mean_out ? a.mean_out + (size_t)bidx * Nq : a.mean + (size_t)bidx * Nq;
    double *oP = a.P_out ? a.P_out + (size_t)bidx * N * N : a.P + (size_t)bidx * N * N;
    if (wave == 1 && lane < N) {
        const int sv = fast_vec_storage(lane);
        if (sv >= 0) omean[sv] = ref[sv];
    }
    if (wave == 0 && lane < NSO3) {
        const int so = lane ? 9 + 7 * lane : 3;
        const Quat qr = ldq(ref + so);
        omean[so] = qr.x; omean[so + 1] = qr.y; omean[so + 2] = qr.z; omean[so + 3] = qr.w;
    }
    if (tid == 0) a.outliers[bidx] = nout;
    SLK_FSTAMP(13);

    // ---- P+ (:665 -> :574-589): p = sum_j e^_j + (N + 1/2) / 2 d0 by waves 2 / 3, the tile columns one per wave
    if (wave >= 2) {
        const int hr = wave - 2;
        double s = 0.0;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) s += Et[(ks * 2 + hr) * 64 + lane];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const int rho = 16 * hr + lane;
        if (lane < 16 && rho < NROT) pd[rho] = s + (0.5 * ((double)N + 0.5)) * pd[32 + rho];
    }
    d4 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    if (wave == 0) fast_rebuild_col<K, 0>(Lt, Et, str, lane, acc);
    else if (wave == 1) fast_rebuild_col<K, 1>(Lt, Et, str, lane, acc);
    else if (wave == 2) fast_rebuild_col<K, 2>(Lt, Et, str, lane, acc);
    else if constexpr (NT > 3) fast_rebuild_col<K, 3>(Lt, Et, str, lane, acc);
    __syncthreads();                                             // factor and E^ are dead; p is complete
    SLK_FSTAMP(14);
    if (wave == 0) fast_store_col<K, 0>(pd, Lt, oP, lane, acc, a.lower_only != 0);
    else if (wave == 1) fast_store_col<K, 1>(pd, Lt, oP, lane, acc, a.lower_only != 0);
    else if (wave == 2) fast_store_col<K, 2>(pd, Lt, oP, lane, acc, a.lower_only != 0);
    else if constexpr (NT > 3) fast_store_col<K, 3>(pd, Lt, oP, lane, acc, a.lower_only != 0);
    SLK_STAMP_NR(15);
    return true;
}

} // namespace slk
