// slk_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the sigma-point Kalman
// hot path of localization::Msckf / localization::Usckf (reference src/filters/Msckf.hpp,
// Usckf.hpp).  One workgroup owns one filter.  Per step HBM sees the lower triangle of P twice
// (second time from L2 / Infinity Cache) and {mean, P} once out; everything in between lives in
// registers and LDS: register-resident Cholesky -> packed factor in LDS -> sigma points ->
// measurement map -> moments -> gain -> downdate fused into the second Cholesky -> manifold mean
// -> fp64 MFMA covariance rebuild.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/slk.h"
#include "slk_math.hpp"

namespace slk {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

constexpr int KP = 8;        // sigma points per rebuild panel (2 MFMA k-steps), double-buffered
#ifndef SLK_WGS
#define SLK_WGS 4     // workgroups per CU the N <= 64 kernels are built for (register budget 512 / SLK_WGS)
#endif
constexpr int MAXM = 32;     // max measurement rows handled on chip
constexpr int PRED_SCRATCH = 1536;  // doubles of pool used by the 12-DOF predict phase

struct Lay { int kind, k, nfk, nfkl, N, Nq, nso3; };

struct KArgs {
    int B;
    Lay lay;
    double *mean; double *P; int *status; unsigned *outliers;
    // predict
    int do_predict, pm; const double *u; int u_stride; const double *Q; int q_stride; const double *Yext;
    // update
    int do_update, mm; const double *mp; int mp_stride; const double *z; int m;
    const double *R; int r_stride; int gate; const double *Zext;
    const int *rowsel;    // gate == 2: [B][m + 2] = surviving rows, outliers, row indices (device)
    // tier B sigma-point emission: 1 = predict sigma points, 2 = update sigma points
    int emit; double *Xout;     // 3 = checkSigmaPoints: re-draw, mean and covariance into mean_out / P_out
    double *mean_out, *P_out;   // null = in place
    int lower_only;             // the fast path stores the lower triangle (and the diagonal tiles) of P+ only: slk_mirror_upper_kernel later
    int rebuild_prec;     // covariance rebuild arithmetic: 0 = fp64 (parity path), 1 = fp32 MFMA, 2 = bf16 inputs / fp32 accumulate
    double *wsL, *wsDR;   // global workspaces: packed factor (large states N > 80; factor hand-off of msckf_chol_kernel), rotation deviations
    int *wsfail;          // msckf_chol_kernel -> step kernel: first non-positive pivot per filter, or -1
    const unsigned long long *rtab;   // Msckf: descriptors of the rotation items (layout only, built by the host)
    long long *dbg;   // phase stamps, diagnostic builds (-DSLK_STAMPS) only; always null in the product
    int stop;         // diagnostic builds: leave the kernel after this stamp (per-phase instruction counts)
};

#ifdef SLK_STAMPS
#define SLK_STAMP(i) do { if (tid == 0 && a.dbg) a.dbg[(size_t)bidx * 32 + (i)] = clock64(); if (a.stop > 0 && a.stop == (i)) return; } while (0)
#define SLK_STAMP_NR(i) do { if (tid == 0 && a.dbg) a.dbg[(size_t)bidx * 32 + (i)] = clock64(); } while (0)
#define SLK_NOTE(i, v) do { if (tid == 0 && a.dbg) a.dbg[(size_t)bidx * 32 + (i)] = (long long)(v); } while (0)
#else
#define SLK_STAMP(i) do { } while (0)
#define SLK_STAMP_NR(i) do { } while (0)
#define SLK_NOTE(i, v) do { } while (0)
#endif

// ------------------------------------------------------------------ layout helpers
// State.hpp:141-149, :246-252, :384-396 (MultiState tangent order), :567-588 (AugmentedState)
__host__ __device__ __forceinline__ int so3_toff(const Lay &L, int b)
{
    return L.kind == SLK_MSCKF ? (b == 0 ? 3 : 12 + 6 * (b - 1) + 3) : 12 * b + 3;
}
__host__ __device__ __forceinline__ int so3_soff(const Lay &L, int b)
{
    return L.kind == SLK_MSCKF ? (b == 0 ? 3 : 13 + 7 * (b - 1) + 3) : 13 * b + 3;
}
// tangent index -> storage index of a vector component, or -1 with (blk, comp) of an SO(3) block
__host__ __device__ __forceinline__ int t2s(const Lay &L, int t, int &blk, int &comp)
{
    if (L.kind == SLK_MSCKF) {
        if (t < 12) {
            if (t < 3) return t;
            if (t < 6) { blk = 0; comp = t - 3; return -1; }
            return t + 1;
        }
        int c = (t - 12) / 6, r = (t - 12) % 6;
        if (r < 3) return 13 + 7 * c + r;
        blk = c + 1; comp = r - 3;
        return -1;
    }
    if (t < 36) {
        int s = t / 12, r = t % 12;
        if (r < 3) return 13 * s + r;
        if (r < 6) { blk = s; comp = r - 3; return -1; }
        return 13 * s + r + 1;
    }
    return 39 + (t - 36);
}
// pose index of a measurement model -> tangent offset, storage offset, SO(3) block
__host__ __device__ __forceinline__ void pose_of(const Lay &L, int c, int &tp, int &sp, int &b)
{
    if (L.kind == SLK_MSCKF) {
        if (c == 0) { tp = 0; sp = 0; b = 0; }
        else { tp = 12 + 6 * (c - 1); sp = 13 + 7 * (c - 1); b = c; }
    } else { tp = 12 * c; sp = 13 * c; b = c; }
}

__host__ __device__ inline int round_up(int x, int q) { return (x + q - 1) / q * q; }

// number of sigma points whose SO(3) block b differs from X_0's: columns j <= toff_b + 2 of the
// lower-triangular factor, two signs, plus i = 0
__host__ __device__ __forceinline__ int rot_count(const Lay &L, int b)
{
    int c = 2 * (so3_toff(L, b) + 3) + 1, S = 2 * L.N + 1;
    return c < S ? c : S;
}

// ------------------------------------------------------------------ packed lower-triangular factor
// column j holds rows j..n-1 contiguously: element (i, j), i >= j, at j*(2n - j + 1)/2 + (i - j)
// (device: 24-bit multiply -- full rate, the indices stay far below 2^23 -- instead of the quarter-rate 32-bit one)
__host__ __device__ __forceinline__ int pkcol(int n, int j)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (__mul24(j, 2 * n - j + 1) >> 1) - j;
#else
    return ((j * (2 * n - j + 1)) >> 1) - j;
#endif
}
__host__ __device__ __forceinline__ int pk(int n, int i, int j) { return pkcol(n, j) + i; }
// column base of the packed factor for column k0 + g (g = lane >> 4, tri_g = g (g + 1) / 2): pk(n, t, k0 + g) = that + t
__device__ __forceinline__ int packed_colbase(int n, int k0, int g, int tri_g)
{
    return ((k0 * (2 * n - k0 - 1)) >> 1) + __mul24(g, n - k0) - tri_g;
}

__host__ __device__ __forceinline__ int pk_size(int n) { return n * (n + 1) / 2; }
__device__ __forceinline__ double Lz(const double *Lp, int n, int t, int j)
{
    // branch-free: always load (index clamped into the array), then select -- lets the loads of one
    // sigma point issue back to back instead of one exec-masked block each
    const bool in = j <= t;
    const double v = Lp[in ? pk(n, t, j) : 0];
    return in ? v : 0.0;
}

// ------------------------------------------------------------------ LDS carve (in doubles)
struct Carve {
    int Lp, mu, ref, delta, md, cq, pn12, small, colbuf, pool, total;
    int S, LDD, TN, W;   // W = rotation-row items that differ from X_0 (sum over blocks of rot_count)
};

// big = large-state variant (NT > 4, i.e. N > 64): the packed factor and the rotation deviations live in a global
// workspace, LDS keeps the small vectors, the measurement arrays, a Cholesky panel and the MFMA panels
// doubles reserved for Z [S][m]; where the factor-update path can run (one tile row per wave, m <= 8) its W buffer
// (BW_SIZE = 512 doubles) takes Z's place after the moments, so the region is at least that large
__host__ __device__ inline int z_region(int S, int m, int NT, bool big)
{
    const int z = round_up(S * m, 2);
    if (big && m <= 8) {             // factor_update_blocks: W [N][8] + wave totals [8][36] after the moments
        const int w = 8 * ((S - 1) / 2) + 8 * 36 + 8;
        return z > w ? z : w;
    }
    return (!big && NT >= 3 && NT <= 4 && m <= 8 && z < 512) ? 512 : z;
}

__host__ __device__ inline Carve carve_step(const Lay &L, int m, int NT, bool big = false, int prec = 0)
{
    const int N = L.N, Nq = L.Nq;
    Carve c;
    c.S = 2 * N + 1;
    c.TN = 16 * NT;
    c.LDD = 16 * NT + ((NT & 1) ? 0 : 16);       // LDD % 32 == 16: the two 16-lane halves of a b64 read hit disjoint banks
    c.W = 0;
    for (int b = 0; b < L.nso3; ++b) c.W += rot_count(L, b);
    int o = 0;
    c.Lp = o;     o += big ? 0 : round_up(pk_size(N), 2);
    c.mu = o;     o += round_up(Nq, 2);
    c.ref = o;    o += round_up(Nq, 2);
    c.delta = o;  o += round_up(N, 2);
    c.md = o;     o += round_up(N, 2);
    c.cq = o;     o += 4 * L.nso3;                 // ref_b^-1 * mu_b per SO(3) block (mean loop)
    c.pn12 = o;   o += (NT <= 2) ? 144 : 0;        // the predicted 12 x 12 block (one-wave kernels: predict runs inside)
    c.small = o;  o += 96;
    c.colbuf = o; o += big ? (4 * 34 + 136 + 16) : 4 * ((c.TN > 32 ? c.TN : 32) + 2);   // big: cholm<1..2> buffer + packed 16x16 factor + its reciprocal pivots
    c.pool = o;
    // measurement part: Z[S*m] (the gain K[N*m] reuses its place once the moments are done) DZ[N*m] Pxz[N*m]
    // Sm[m*m] G[m*(2m+1)] zbar innov
    int upd1 = z_region(c.S, m, NT, big) + 2 * round_up(N * m, 2) + round_up(m * m, 2) + round_up(m * (2 * m + 1), 2)
               + 4 * round_up(m, 2);
    // applyDelta part: rotation deviations (3 per item that differs from X_0) + the double-buffered panels of the
    // panel rebuild (N > 64, or the reduced-precision sweep); the K-split rebuild stages nothing
    const bool panels = big || NT > 4 || prec != 0;
    int upd2 = (big ? 0 : round_up(3 * c.W, 2)) + (panels ? 2 * KP * c.LDD : 0);
    int pool = PRED_SCRATCH;
    if (upd1 > pool) pool = upd1;
    if (upd2 > pool) pool = upd2;
    if (big && upd1 + c.TN * 17 > pool) pool = upd1 + c.TN * 17;       // Cholesky panel next to Pxz / K (downdate reads them)
    c.total = o + pool;
    return c;
}

// ------------------------------------------------------------------ implicit sigma points
// generateSigmaPoints (Msckf.hpp:407-431 / :442-468): X0 = mu + delta, X(2j+1) = mu + (delta + L.col(j)),
// X(2j+2) = mu + (delta - L.col(j)); L = packed lower factor.
struct Sig { int j; double sgn; };
__device__ __forceinline__ Sig sig_of(int i)
{
    Sig s;
    s.j = (i > 0) ? ((i - 1) >> 1) : 0;
    s.sgn = (i == 0) ? 0.0 : ((i & 1) ? 1.0 : -1.0);
    return s;
}
__device__ __forceinline__ double pert(const double *Lp, int n, const double *delta, int t, const Sig &s)
{
    double l = s.sgn * Lz(Lp, n, t, s.j);       // sgn = 0 for X_0
    return delta ? (delta[t] + l) : l;
}
__device__ __forceinline__ Quat sigma_quat(const Lay &L, const double *mu, const double *Lp,
                                           const double *delta, int b, const Sig &s)
{
    int to = so3_toff(L, b);
    return qmul(ldq(mu + so3_soff(L, b)),
                so3_exp(pert(Lp, L.N, delta, to, s), pert(Lp, L.N, delta, to + 1, s), pert(Lp, L.N, delta, to + 2, s)));
}

// ------------------------------------------------------------------ Msckf rotation items
// MultiState layout in closed form (State.hpp:384-396): SO(3) block b sits at tangent 3 / 9 + 6b and storage
// 3 / 9 + 7b (b = 0 / b >= 1); rot_count(b) = 13 / 25 + 12b sigma points differ from X_0 in that block, so the
// prefix offsets of the flattened (block, sigma point) items are roff(b) = 6b^2 + 19b - 12 (b >= 1).
__host__ __device__ __forceinline__ int msckf_toff(int b) { return b ? 9 + 6 * b : 3; }
__host__ __device__ __forceinline__ int msckf_soff(int b) { return b ? 9 + 7 * b : 3; }
__host__ __device__ __forceinline__ int msckf_roff(int b) { return b ? 6 * b * b + 19 * b - 12 : 0; }
// item w -> (block, index in block) without a table: invert the quadratic, repair the float rounding
__device__ __forceinline__ void msckf_rot_item(int w, int &b, int &i)
{
    int bb = 0;
    if (w >= 13) {
        bb = (int)((__builtin_amdgcn_sqrtf((float)(649 + 24 * w)) - 19.0f) * (1.0f / 12.0f));   // raw v_sqrt_f32, repaired below
        if (msckf_roff(bb + 1) <= w) ++bb;
        if (msckf_roff(bb) > w) --bb;
    }
    b = bb;
    i = w - msckf_roff(bb);
}
// The (block, sigma point) -> addresses / signs arithmetic depends on the layout only: the host tabulates it
// once per handle (rot_item_descriptor), the kernels unpack one 64-bit word per item.
//   bits 0-15 index of L(toff+2, j) in the packed factor, 16 / 17 "j <= toff" / "j <= toff+1",
//   18-19 sign (0: X_0, 1: +L_j, 2: -L_j), 20-27 toff, 28-37 soff, 38-43 SO(3) block
__host__ __device__ inline unsigned long long rot_item_descriptor(int N, int w)
{
    int b = 0;
    while (msckf_roff(b + 1) <= w) ++b;
    const int i = w - msckf_roff(b), to = msckf_toff(b), so = msckf_soff(b);
    const int j = (i > 0) ? ((i - 1) >> 1) : 0;
    const unsigned long long a2 = (unsigned long long)(pk(N, to + 2, j));
    const unsigned long long sc = (i == 0) ? 0 : ((i & 1) ? 1 : 2);
    return a2 | ((unsigned long long)(j <= to) << 16) | ((unsigned long long)(j <= to + 1) << 17) | (sc << 18)
           | ((unsigned long long)to << 20) | ((unsigned long long)so << 28) | ((unsigned long long)b << 38);
}
// cq[4 b] = ref_b^-1 * mu_b per SO(3) block (kept current by whoever moves the reference): (mu_b exp(v)) [-] ref_b =
// log(ref_b^-1 mu_b exp(v)) = log(cq_b exp(v)) -- one quaternion product per item instead of two
__device__ __forceinline__ void rot_deviation_desc(unsigned long long d, const double *cq, const double *Lp,
                                                   const double *delta, double &dx, double &dy, double &dz)
{
    const unsigned lo = (unsigned)d;
    const int a2 = lo & 0xffff, to = (lo >> 20) & 0xff, b = (int)((d >> 38) & 0x3f);
    const bool in0 = lo & (1u << 16), in1 = lo & (1u << 17);
    const unsigned sc = (lo >> 18) & 3;
    const double sgn = (sc == 1) ? 1.0 : ((sc == 2) ? -1.0 : 0.0);
    const double l0 = Lp[in0 ? a2 - 2 : 0], l1 = Lp[in1 ? a2 - 1 : 0], l2 = Lp[a2];
    const double d0 = delta[to], d1 = delta[to + 1], d2 = delta[to + 2];
    const Quat c = ldq(cq + 4 * b);
    const double v0 = d0 + sgn * (in0 ? l0 : 0.0), v1 = d1 + sgn * (in1 ? l1 : 0.0), v2 = d2 + sgn * l2;
    so3_log(qmul(c, so3_exp(v0, v1, v2)), dx, dy, dz);
}
// (mu [+] (delta +- L_j))_b [-] ref_b for item (b, i): all operands loaded up front (clamped addresses,
// selects afterwards) so the LDS round trips overlap; an item exists only for j <= toff + 2
__device__ __forceinline__ void rot_deviation(const double *mu, const double *ref, const double *Lp, const double *delta,
                                              int N, int b, int i, double &dx, double &dy, double &dz)
{
    const int to = msckf_toff(b), so = msckf_soff(b);
    const int j = (i > 0) ? ((i - 1) >> 1) : 0;
    const double sgn = (i == 0) ? 0.0 : ((i & 1) ? 1.0 : -1.0);
    const int jb = pkcol(N, j);                                       // pk(N, t, j) = jb + t
    const bool in0 = j <= to, in1 = j <= to + 1;
    const double l0 = Lp[in0 ? jb + to : 0], l1 = Lp[in1 ? jb + to + 1 : 0], l2 = Lp[jb + to + 2];
    const double d0 = delta[to], d1 = delta[to + 1], d2 = delta[to + 2];
    const Quat qm = ldq(mu + so), qr = ldq(ref + so);
    const double v0 = d0 + sgn * (in0 ? l0 : 0.0), v1 = d1 + sgn * (in1 ? l1 : 0.0), v2 = d2 + sgn * l2;
    so3_boxminus(qmul(qm, so3_exp(v0, v1, v2)), qr, dx, dy, dz);
}

// ------------------------------------------------------------------ register-resident Cholesky
// Lower Cholesky of an n x n matrix with the trailing matrix held in REGISTERS, block-cyclic over a
// GD x GD thread grid (thread (ti,tj) owns elements i = ti + GD*sa, j = tj + GD*sb).  Per column: the
// owners publish the raw column through a double-buffered LDS vector, ONE barrier, everybody applies
// a_ij -= c_i c_j / d.  The pivot uses v_rsq_f64 + two Newton steps instead of sqrt and two divisions
// (they were the critical path).  The finished factor goes to the packed array Lp.  init(i, j)
// supplies the initial lower-triangle element (global memory, or P minus the gain downdate), so no
// copy of P is staged in LDS.  Returns -1 or the first non-positive pivot (same in every thread).
// Eigen::LLT in the reference never has its info() read (Msckf.hpp:412-413, Usckf.hpp:537-538).
template <int NTHREADS> struct Grid { static constexpr int GD = (NTHREADS >= 256) ? 16 : 8; };

__device__ __forceinline__ void rsqrt_pivot(double d, double &sq, double &rs)
{
    double y = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    sq = d * y;                             // sqrt(d): y is 1/sqrt(d) to about an ulp already
    rs = y;
}

// Two pivot columns per barrier (rank-2 step): columns k and k+1 are published raw, every thread
// redoes the tiny 2x2 pivot algebra itself, then a_ij -= l0_i l0_j + l1_i l1_j.  Slots with i <= k+1 or
// j <= k+1 are dead after the step, so no masking of the LDS reads is needed (garbage only ever
// reaches dead slots); an odd n is padded with a unit diagonal element by chol_packed.
template <int NTHREADS, int SD, int KB>
struct CholCols {
    __device__ __forceinline__ static void run(double (&a)[SD][SD], double *Lp, int n, double *colbuf,
                                               int ti, int tj, bool active, int &fail)
    {
        if constexpr (KB < SD) {
            constexpr int GD = Grid<NTHREADS>::GD;
            constexpr int CB = SD * GD;
            if (fail < 0) {
                for (int kt = 0; kt < GD; kt += 2) {
                    const int k = KB * GD + kt;
                    if (k >= n) break;
                    double *buf0 = colbuf + ((k >> 1) & 1) * (2 * CB), *buf1 = buf0 + CB;
                    if (active && (tj == kt || tj == kt + 1)) {
                        double *bw = (tj == kt) ? buf0 : buf1;
#pragma unroll
                        for (int sa = 0; sa < SD; ++sa) bw[ti + GD * sa] = a[sa][KB];
                    }
                    __syncthreads();
                    const double d0 = buf0[k], e01 = buf0[k + 1], d1raw = buf1[k + 1];
                    double li0[SD], li1[SD], lj0[SD], lj1[SD];
#pragma unroll
                    for (int sa = 0; sa < SD; ++sa) { li0[sa] = buf0[ti + GD * sa]; li1[sa] = buf1[ti + GD * sa]; }
#pragma unroll
                    for (int sb = KB; sb < SD; ++sb) { lj0[sb] = buf0[tj + GD * sb]; lj1[sb] = buf1[tj + GD * sb]; }
                    if (!(d0 > 0.0)) { fail = k; break; }
                    double sq0, r0, sq1, r1;
                    rsqrt_pivot(d0, sq0, r0);
                    const double l10 = e01 * r0;
                    const double d1 = fma(-l10, l10, d1raw);
                    if (!(d1 > 0.0)) { fail = k + 1; break; }
                    rsqrt_pivot(d1, sq1, r1);
#pragma unroll
                    for (int sa = 0; sa < SD; ++sa) { li0[sa] *= r0; li1[sa] = fma(-li0[sa], l10, li1[sa]) * r1; }
#pragma unroll
                    for (int sb = KB; sb < SD; ++sb) { lj0[sb] *= r0; lj1[sb] = fma(-lj0[sb], l10, lj1[sb]) * r1; }
                    if (active && (tj == kt || tj == kt + 1)) {
                        const bool first = (tj == kt);
                        const int kc = first ? k : k + 1;
                        if (kc < n) {
                            const int base = pk(n, kc, kc);
                            const double sq = first ? sq0 : sq1;
#pragma unroll
                            for (int sa = 0; sa < SD; ++sa) {
                                int i = ti + GD * sa;
                                double v = (i == kc) ? sq : (first ? li0[sa] : li1[sa]);
                                if (i >= kc && i < n) Lp[base + (i - kc)] = v;
                            }
                        }
                    }
#pragma unroll
                    for (int sa = 0; sa < SD; ++sa)
#pragma unroll
                        for (int sb = KB; sb < SD; ++sb) a[sa][sb] -= fma(li0[sa], lj0[sb], li1[sa] * lj1[sb]);
                }
            }
            CholCols<NTHREADS, SD, KB + 1>::run(a, Lp, n, colbuf, ti, tj, active, fail);
        }
    }
};

// colbuf needs 4 * SD * GD doubles
template <int NTHREADS, int SD, class InitFn>
__device__ __forceinline__ int chol_packed(double *Lp, int n, double *colbuf, int tid, InitFn init)
{
    constexpr int GD = Grid<NTHREADS>::GD;
    const int ti = tid % GD, tj = tid / GD;
    const bool active = tid < GD * GD;
    double a[SD][SD];
#pragma unroll
    for (int sa = 0; sa < SD; ++sa)
#pragma unroll
        for (int sb = 0; sb < SD; ++sb) {
            int i = ti + GD * sa, j = tj + GD * sb;
            a[sa][sb] = (active && i < n && j <= i) ? init(i, j) : ((i == n && j == n) ? 1.0 : 0.0);
        }
    int fail = -1;
    CholCols<NTHREADS, SD, 0>::run(a, Lp, n, colbuf, ti, tj, active, fail);
    __syncthreads();
    return (fail >= n) ? -1 : fail;
}

// ------------------------------------------------------------------ single-wave blocked Cholesky on the matrix cores
// The whole lower triangle lives in ONE wave as v_mfma_f64_16x16x4 accumulator tiles (lane l, reg r of
// tile (I,J): row 16I + (l>>4) + 4r, col 16J + (l&15)).  A step retires FOUR columns: the 16 lanes
// holding them publish the raw columns to LDS, every lane redoes the 4x4 pivot block itself (four
// v_rsq_f64 pivots), forward-substitutes the panel rows it needs as its A/B fragment (lane l ->
// L[16I + (l&15)][k0 + (l>>4)], which is also the element it stores to the packed factor) and the
// trailing matrix gets acc(I,J) -= L_I L_J^T as one MFMA per tile (rank-4 update).  No workgroup
// barrier: LDS traffic of one wave is ordered.  15 steps for n = 60 instead of 60 column steps.
__host__ __device__ constexpr int tile_idx(int I, int J) { return I * (I + 1) / 2 + J; }

__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int NT> struct CholM {
    static constexpr int NTL = NT * (NT + 1) / 2;
    static constexpr int LDC = 16 * NT + 2;
    static constexpr int COLBUF = 4 * LDC;
};

template <int NT, class InitFn>
__device__ __forceinline__ void cholm_load(d4 (&acc)[CholM<NT>::NTL], int n, int lane, InitFn init)
{
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = 16 * I + g + 4 * r, col = 16 * J + c;
                double v;
                if (row < n && col < n) v = (row >= col) ? init(row, col) : init(col, row);
                else v = (row == col) ? 1.0 : 0.0;
                acc[tile_idx(I, J)][r] = -v;          // the accumulators hold -A: the rank-4 updates ADD L L^T
            }
}

// The same tiles TRANSPOSED (cholp_factor): lane l, reg r of tile (I, J): row 16I + (l&15), col 16J + (l>>4) + 4r -- sixteen
// lanes run down a column of the matrix (128 contiguous bytes per request instead of sixteen 32-byte pieces), and the four
// pivot columns of a step sit in ONE register of every lane, already in the fragment distribution.
template <int NT, class InitFn>
__device__ __forceinline__ void cholm_load_t(d4 (&acc)[CholM<NT>::NTL], int n, int lane, InitFn init)
{
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int I = J; I < NT; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = 16 * I + c, col = 16 * J + g + 4 * r;
                double v;
                if (row < n && col < n) v = (row >= col) ? init(row, col) : init(col, row);
                else v = (row == col) ? 1.0 : 0.0;
                acc[tile_idx(I, J)][r] = -v;          // the accumulators hold -A: the rank-4 updates ADD L L^T
            }
}

// acc -= X * Y^T with X, Y n x kk column-major panels in LDS (ld ldx / ldy); column c of X is xcol(c)
template <int NT, class XFn, class YFn>
__device__ __forceinline__ void cholm_downdate(d4 (&acc)[CholM<NT>::NTL], int n, int kk, int lane, XFn xel, YFn yel)
{
    const int c = lane & 15, g = lane >> 4;
    for (int ks = 0; ks * 4 < kk; ++ks) {
        const int cc = 4 * ks + g;
        double af[NT], bf[NT];
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            int rho = 16 * I + c;
            bool ok = cc < kk && rho < n;
            af[I] = ok ? xel(rho, cc) : 0.0;           // (-A) += X Y^T
            bf[I] = ok ? yel(rho, cc) : 0.0;
        }
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J)
                acc[tile_idx(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[I], bf[J], acc[tile_idx(I, J)], 0, 0, 0);
    }
}

template <int NT, int JK>
struct CholMCols {
    __device__ __forceinline__ static void run(d4 (&acc)[CholM<NT>::NTL], double *Lp, int n, double *colbuf, int lane, int &fail)
    {
        if constexpr (JK < NT) {
            constexpr int LDC = CholM<NT>::LDC;
            const int c = lane & 15, g = lane >> 4;
            if (fail < 0) {
                for (int c0 = 0; c0 < 16; c0 += 4) {
                    const int k0 = 16 * JK + c0;
                    if (k0 >= n) break;
                    // 1. publish the four pivot columns, raw
                    if (c >= c0 && c < c0 + 4) {
                        double *dst = colbuf + (c - c0) * LDC + g;
#pragma unroll
                        for (int I = JK; I < NT; ++I)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dst[16 * I + 4 * r] = acc[tile_idx(I, JK)][r];
                    }
                    wave_sync();
                    // 2. 4x4 pivot block (same values in every lane) and this lane's raw panel rows
                    const double *pb = colbuf + k0;
                    const double p00 = -pb[0], p10 = -pb[1], p20 = -pb[2], p30 = -pb[3];          // published values are -A
                    const double p11 = -pb[LDC + 1], p21 = -pb[LDC + 2], p31 = -pb[LDC + 3];
                    const double p22 = -pb[2 * LDC + 2], p32 = -pb[2 * LDC + 3], p33 = -pb[3 * LDC + 3];
                    double v[NT][4];
#pragma unroll
                    for (int I = JK; I < NT; ++I)
#pragma unroll
                        for (int b = 0; b < 4; ++b) v[I][b] = -colbuf[b * LDC + 16 * I + c];
                    wave_sync();       // the next step's publish must not overtake these reads
                    double s0, r0, s1, r1, s2, r2, s3, r3;
                    if (!(p00 > 0.0)) { fail = k0; break; }
                    rsqrt_pivot(p00, s0, r0);
                    const double l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
                    const double d1 = fma(-l10, l10, p11);
                    if (!(d1 > 0.0)) { fail = k0 + 1; break; }
                    rsqrt_pivot(d1, s1, r1);
                    const double l21 = fma(-l20, l10, p21) * r1, l31 = fma(-l30, l10, p31) * r1;
                    const double d2 = fma(-l21, l21, fma(-l20, l20, p22));
                    if (!(d2 > 0.0)) { fail = k0 + 2; break; }
                    rsqrt_pivot(d2, s2, r2);
                    const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * r2;
                    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33)));
                    if (!(d3 > 0.0)) { fail = k0 + 3; break; }
                    rsqrt_pivot(d3, s3, r3);
                    // 3. forward substitution of the panel rows -> fragments = factor entries
                    const int kap = k0 + g;
                    const double sg = (g == 0) ? s0 : (g == 1) ? s1 : (g == 2) ? s2 : s3;
                    double frag[NT];
#pragma unroll
                    for (int I = JK; I < NT; ++I) {
                        const double vi0 = v[I][0], vi1 = v[I][1], vi2 = v[I][2], vi3 = v[I][3];
                        const double x0 = vi0 * r0;
                        const double x1 = fma(-x0, l10, vi1) * r1;
                        const double x2 = fma(-x1, l21, fma(-x0, l20, vi2)) * r2;
                        const double x3 = fma(-x2, l32, fma(-x1, l31, fma(-x0, l30, vi3))) * r3;
                        double f = (g == 0) ? x0 : (g == 1) ? x1 : (g == 2) ? x2 : x3;
                        const int rho = 16 * I + c;
                        if (rho == kap) f = sg;
                        if (rho < kap) f = 0.0;           // strictly upper part of the pivot block / retired rows
                        frag[I] = f;
                        if (rho >= kap && rho < n && kap < n) Lp[pk(n, rho, kap)] = f;
                    }
                    // 4. rank-4 update of the trailing tiles, tile column by tile column: the tiles the NEXT
                    // step publishes (column JK, then JK+1) retire first, the rest drains under its prologue
#pragma unroll
                    for (int J = JK; J < NT; ++J)
#pragma unroll
                        for (int I = J; I < NT; ++I)
                            acc[tile_idx(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[I], frag[J], acc[tile_idx(I, J)], 0, 0, 0);
                }
            }
            CholMCols<NT, JK + 1>::run(acc, Lp, n, colbuf, lane, fail);
        }
    }
};

// factor the matrix held in `acc` (see cholm_load); wave-local, returns -1 or the failing pivot
template <int NT>
__device__ __forceinline__ int cholm_factor(d4 (&acc)[CholM<NT>::NTL], double *Lp, int n, double *colbuf, int lane)
{
    int fail = -1;
    // this wave runs a long dependent chain while the co-resident waves do throughput work: let it win issue
    __builtin_amdgcn_s_setprio(3);
    CholMCols<NT, 0>::run(acc, Lp, n, colbuf, lane, fail);
    wave_sync();
    __builtin_amdgcn_s_setprio(0);
    return fail;
}

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// ------------------------------------------------------------------ the same blocked factorisation, panel by rows
// cholm_factor's step has every lane redo the 4 x 4 pivot block and forward-substitute four columns for each of its
// tile rows, then pick one of the four results: ~250 vector instructions per step, and one wave per filter runs at
// the rate of its instruction count (msckf_chol_kernel: 3.9 k per filter, 46 us per 4096 filters).  Here the four
// published columns are read back with lane = ROW (one value per column), the 64 x 4 panel is factored once for all
// rows -- per column: pivot by v_readlane, v_rsq_f64 + two Newton steps, one scale, one multiply-add per remaining
// panel column -- written to LDS again and fetched as the matrix-core fragments of the rank-4 update; the packed factor
// is stored straight from the row layout.  A non-positive pivot is recorded; what follows it is never used.
// OUT: where the factor goes -- 0: packed, LDS; 1: packed, global memory; 2: the 16 x 16 tiles of the exact-shape update kernel
// (slk_step_fast.hpp: tile (I, J) at (I (I + 1) / 2 + J) * 256, element (t, j) at (j & 15) * 16 + (t & 15), exact zeros above the
// diagonal and in the padding), LDS
template <int NT, int JK, int C0, int OUT = 1>
struct CholPSteps {
    __device__ __forceinline__ static void run(d4 (&acc)[CholM<NT>::NTL], double *Lp, int n, double *colbuf, int lane, bool &bad)
    {
        if constexpr (JK < NT) {
            constexpr int LDC = CholM<NT>::LDC;
            constexpr int k0 = 16 * JK + C0;
            if (OUT == 2 || k0 < n) {
                const int c = lane & 15, g = lane >> 4;
                const int row = (NT == 4) ? lane : min(lane, 16 * NT - 1);
                // 1. publish the four pivot columns, raw (the accumulators hold -A, tiles transposed: cholm_load_t): register
                // C0 / 4 of lane (g, c) is element (16 I + c, k0 + g)
#pragma unroll
                for (int I = JK; I < NT; ++I) colbuf[g * LDC + 16 * I + c] = acc[tile_idx(I, JK)][C0 / 4];
                wave_sync();
                // 2. this lane's row of the panel, and the four columns
                double l[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) l[p] = -colbuf[p * LDC + row];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const double d = readlane_f64(l[p], k0 + p);
                    {   // (tested here and now: left alone the compiler keeps all sixty pivots for one test at the end -- in scalar
                        // registers it does not have, i.e. spilled lane by lane; the flag goes through a vector register)
                        int nb = !(d > 0.0);
                        asm volatile("" : "+v"(nb));
                        bad |= (nb != 0);
                    }
                    double sq, rs;
                    rsqrt_pivot(d, sq, rs);
                    (void)sq;
                    l[p] *= rs;                                   // (lane k0 + p held the pivot: d * rs = sqrt(d))
#pragma unroll
                    for (int q = p + 1; q < 4; ++q) l[q] = fma(-l[p], readlane_f64(l[p], k0 + q), l[q]);
                }
                // (rows above the diagonal of the pivot block and retired rows carry garbage: as fragments they only reach
                // accumulator slots that are dead after this step, and the packed factor takes rows >= column only)
                // 3. the factor panel: to LDS for the fragments, to the packed factor from the row layout
                wave_sync();
                if (NT == 4 || lane < 16 * NT) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) colbuf[p * LDC + lane] = l[p];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    // (the column's base pinned in scalar registers: the store is `global_store v_lane8, data, s[base]` -- left to
                    // itself the compiler keeps ONE base and adds every column's offset with 64-bit vector arithmetic)
                    if constexpr (OUT == 2) {
                        if (lane >= 16 * JK && (NT == 4 || lane < 16 * NT)) {
                            const int It = lane >> 4;
                            Lp[(It * (It + 1) / 2 + JK) * 256 + (C0 + p) * 16 + (lane & 15)] = (k0 + p < n && lane >= k0 + p) ? l[p] : 0.0;
                        }
                    } else if constexpr (OUT == 1) {
                        typedef __attribute__((address_space(1))) double gdouble;
                        unsigned long long colb = reinterpret_cast<unsigned long long>(Lp + pkcol(n, k0 + p));
                        asm volatile("" : "+s"(colb));
                        gdouble *colp = reinterpret_cast<gdouble *>(colb);
                        if (k0 + p < n && lane >= k0 + p && lane < n) colp[lane] = l[p];
                    } else {
                        if (k0 + p < n && lane >= k0 + p && lane < n) Lp[pkcol(n, k0 + p) + lane] = l[p];
                    }
                }
                wave_sync();
                double frag[NT];
#pragma unroll
                for (int I = JK; I < NT; ++I) frag[I] = colbuf[g * LDC + 16 * I + c];
                wave_sync();       // the next step's publish must not overtake these reads
                // 4. rank-4 update of the trailing tiles (the tiles the next step publishes first)
#pragma unroll
                for (int J = JK; J < NT; ++J)
#pragma unroll
                    for (int I = J; I < NT; ++I)
                        acc[tile_idx(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[J], frag[I], acc[tile_idx(I, J)], 0, 0, 0);   // (tile^T += L_J L_I^T)
            }
            CholPSteps<NT, (C0 == 12 ? JK + 1 : JK), (C0 + 4) & 15, OUT>::run(acc, Lp, n, colbuf, lane, bad);
        }
    }
};

// factor the matrix held in `acc` (see cholm_load_t: TRANSPOSED tiles); wave-local, returns -1 or 0 (some pivot was not positive)
template <int NT, int OUT = 1>
__device__ __forceinline__ int cholp_factor(d4 (&acc)[CholM<NT>::NTL], double *Lp, int n, double *colbuf, int lane)
{
    bool bad = false;
    CholPSteps<NT, 0, 0, OUT>::run(acc, Lp, n, colbuf, lane, bad);
    wave_sync();
    return bad ? 0 : -1;
}

// ------------------------------------------------------------------ one-wave register Cholesky of a small matrix
// lane = row, the row's NMAX columns in registers (n <= NMAX <= 32 rows live).  Per column: the pivot by v_readlane, its
// inverse square root (v_rsq_f64 + two Newton steps), one scale, and per trailing column one broadcast (v_readlane) and
// one multiply-add -- a single wave runs at the rate of its instruction count (an fp64 operation issues every ~10.6
// cycles whether or not it depends on the one before: profiles/r01_mfma_valu_overlap.log), so the count is what is
// minimised.  No LDS round trip, no branch: a non-positive pivot is recorded (what follows it is never used).
// 12 x 12: 6.1 k -> 4.6 k cycles including the load of the matrix (profiles/r03_ab_small_state_register_cholesky.log).
// nst < n: the matrix is padded to n x n with a unit diagonal by the caller, the leading nst x nst part of the factor is stored
// (packed for nst).
template <int NMAX, class InitFn>
__device__ __forceinline__ int chol_rows(double *Lp, int n, int lane, InitFn init, int nst = -1)
{
    if (nst < 0) nst = n;
    static_assert(NMAX <= 32, "chol_rows keeps a row per lane in registers");
    double a[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        // every lane fetches its whole row (lanes beyond n: row n - 1): what sits above the diagonal is never broadcast
        // or stored, and neither a predicate nor an address per column has to stay alive
        a[j] = init(min(lane, n - 1), j < n ? j : 0);
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (whatever init broadcasts: four columns' worth at a time)
    }
    bool bad = false;
    // (the column loop is ONE basic block -- the factor is stored after it: with a predicated store per column the
    // compiler sinks the trailing updates to the column that needs them and keeps every broadcast alive in between)
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        if (j < n) {
            const double d = readlane_f64(a[j], j);
            bad |= !(d > 0.0);
            double sq, rs;
            rsqrt_pivot(d, sq, rs);
            (void)sq;
            a[j] *= rs;                               // (lane j held the pivot itself: d * rs = sqrt(d) as rsqrt_pivot forms it)
#pragma unroll
            for (int k = j + 1; k < NMAX; ++k) a[k] = fma(-a[j], readlane_f64(a[j], k), a[k]);
            __builtin_amdgcn_sched_barrier(0);       // one column's broadcasts (scalar registers) at a time
        }
    }
    if (lane < nst) {
#pragma unroll
        for (int j = 0; j < NMAX; ++j)
            if (j < nst && lane >= j) Lp[pk(nst, lane, j)] = a[j];
    }
    wave_sync();
    return bad ? 0 : -1;      // (callers only test the sign)
}

// ------------------------------------------------------------------ the same factorisation over NT waves
// Wave I owns tile row I (tiles (I, 0..I) in acc[0..I], holding -A like cholm_load): per step every wave of the
// trailing part publishes its raw panel rows, redoes the 4x4 pivot block, forward-substitutes ONLY its own 16 rows,
// stores them to the packed factor, and after a second barrier reads the fragments of the tile rows above it for
// its rank-4 updates.  32 accumulator registers per wave instead of 80 (this is what lets a fourth workgroup
// share the CU); two workgroup barriers per step; no early exit on a non-positive pivot (the first one is
// recorded, the NaNs that follow are never used: the caller leaves the filter unchanged).
template <int NT, class InitFn>
__device__ __forceinline__ void cholw_load(d4 (&acc)[NT], int n, int lane, int wave, InitFn init)
{
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = 16 * wave + g + 4 * r, col = 16 * J + c;
            double v = 0.0;
            if (J <= wave && wave < NT) {
                if (row < n && col < n) v = (row >= col) ? init(row, col) : init(col, row);
                else v = (row == col) ? 1.0 : 0.0;
            }
            acc[J][r] = -v;
        }
}

template <int NT, class XFn, class YFn>
__device__ __forceinline__ void cholw_downdate(d4 (&acc)[NT], int n, int kk, int lane, int wave, XFn xel, YFn yel)
{
    const int c = lane & 15, g = lane >> 4;
    if (wave >= NT) return;
    for (int ks = 0; ks * 4 < kk; ++ks) {
        const int cc = 4 * ks + g;
        double bf[NT];
        const int rw = 16 * wave + c;
        const bool okw = cc < kk && rw < n;
        const double af = okw ? xel(rw, cc) : 0.0;           // (-A) += X Y^T
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            int rho = 16 * J + c;
            bool ok = cc < kk && rho < n && J <= wave;
            bf[J] = ok ? yel(rho, cc) : 0.0;
        }
#pragma unroll
        for (int J = 0; J < NT; ++J)
            if (J <= wave) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf[J], acc[J], 0, 0, 0);
    }
}

// coef: 16 doubles of scratch -- the wave that owns the pivot block does its 4x4 algebra alone (wave-local LDS round
// trip before the first barrier) and publishes r0..r3, s0..s3, l10 l20 l30 l21 l31 l32; the other waves read them
template <int NT>
__device__ __forceinline__ int cholw_factor(d4 (&acc)[NT], double *Lp, int n, double *colbuf, double *coef, int lane,
                                            int wave, int *flag)
{
    constexpr int LDC = CholM<NT>::LDC;
    const int c = lane & 15, g = lane >> 4, tri_g = (g * (g + 1)) >> 1;
    int fail = -1;
    if (wave == 0 && lane == 0) *flag = 0x7fffffff;      // ordered before the atomicMin at the end by the step barriers
#pragma unroll
    for (int JK = 0; JK < NT; ++JK) {
        const bool part = wave >= JK && wave < NT;
#pragma unroll
        for (int c0 = 0; c0 < 16; c0 += 4) {
            const int k0 = 16 * JK + c0;
            if (k0 < n) {                                    // uniform over the workgroup
                // 1. publish the raw panel rows of this wave (four columns)
                if (part && c >= c0 && c < c0 + 4) {
                    double *dst = colbuf + (c - c0) * LDC + 16 * wave + g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[4 * r] = acc[JK][r];
                }
                if (wave == JK) {
                    // 2a. the owner of the pivot block does the 4x4 algebra for everybody (published values are -A)
                    wave_sync();
                    const double *pb = colbuf + k0;
                    const double p00 = -pb[0], p10 = -pb[1], p20 = -pb[2], p30 = -pb[3];
                    const double p11 = -pb[LDC + 1], p21 = -pb[LDC + 2], p31 = -pb[LDC + 3];
                    const double p22 = -pb[2 * LDC + 2], p32 = -pb[2 * LDC + 3], p33 = -pb[3 * LDC + 3];
                    double s0, r0, s1, r1, s2, r2, s3, r3;
                    if (fail < 0 && !(p00 > 0.0)) fail = k0;
                    rsqrt_pivot(p00, s0, r0);
                    const double l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
                    const double d1 = fma(-l10, l10, p11);
                    if (fail < 0 && !(d1 > 0.0)) fail = k0 + 1;
                    rsqrt_pivot(d1, s1, r1);
                    const double l21 = fma(-l20, l10, p21) * r1, l31 = fma(-l30, l10, p31) * r1;
                    const double d2 = fma(-l21, l21, fma(-l20, l20, p22));
                    if (fail < 0 && !(d2 > 0.0)) fail = k0 + 2;
                    rsqrt_pivot(d2, s2, r2);
                    const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * r2;
                    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33)));
                    if (fail < 0 && !(d3 > 0.0)) fail = k0 + 3;
                    rsqrt_pivot(d3, s3, r3);
                    if (lane == 0) {
                        coef[0] = r0; coef[1] = r1; coef[2] = r2; coef[3] = r3;
                        coef[4] = s0; coef[5] = s1; coef[6] = s2; coef[7] = s3;
                        coef[8] = l10; coef[9] = l20; coef[10] = l30; coef[11] = l21; coef[12] = l31; coef[13] = l32;
                    }
                }
                __syncthreads();
                double frag = 0.0;
                if (part) {
                    // 2b. the pivot coefficients and this lane's raw panel row
                    const double r0 = coef[0], r1 = coef[1], r2 = coef[2], r3 = coef[3];
                    const double s0 = coef[4], s1 = coef[5], s2 = coef[6], s3 = coef[7];
                    const double l10 = coef[8], l20 = coef[9], l30 = coef[10], l21 = coef[11], l31 = coef[12], l32 = coef[13];
                    double v[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) v[b] = -colbuf[b * LDC + 16 * wave + c];
                    // 3. forward substitution of this wave's rows -> fragment = factor entries
                    const int kap = k0 + g;
                    const double sg = (g == 0) ? s0 : (g == 1) ? s1 : (g == 2) ? s2 : s3;
                    const double x0 = v[0] * r0;
                    const double x1 = fma(-x0, l10, v[1]) * r1;
                    const double x2 = fma(-x1, l21, fma(-x0, l20, v[2])) * r2;
                    const double x3 = fma(-x2, l32, fma(-x1, l31, fma(-x0, l30, v[3]))) * r3;
                    double f = (g == 0) ? x0 : (g == 1) ? x1 : (g == 2) ? x2 : x3;
                    const int rho = 16 * wave + c;
                    if (rho == kap) f = sg;
                    if (rho < kap) f = 0.0;               // strictly upper part of the pivot block
                    frag = f;
                    if (rho >= kap && rho < n && kap < n) Lp[packed_colbase(n, k0, g, tri_g) + rho] = f;
                }
                __syncthreads();
                if (part) {
                    // 4. rank-4 update of this wave's tiles; fragments of the tile rows above come from the factor
                    const int kap = k0 + g;
                    const int cb = packed_colbase(n, k0, g, tri_g) + c;      // one column base per step, tile rows by offset
                    double fj[NT];
#pragma unroll
                    for (int J = JK; J < NT; ++J) {
                        const int rho = 16 * J + c;
                        const bool in = J < wave && rho >= kap && rho < n && kap < n;
                        const double lv = Lp[in ? cb + 16 * J : 0];
                        fj[J] = (J == wave) ? frag : (in ? lv : 0.0);
                    }
#pragma unroll
                    for (int J = JK; J < NT; ++J)
                        if (J <= wave) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag, fj[J], acc[J], 0, 0, 0);
                }
            }
        }
    }
    // every wave owns some pivot blocks: the first failing pivot over all of them (flag preset to -1 by the caller's
    // barrier-separated write below)
    if (lane == 0 && fail >= 0) atomicMin(reinterpret_cast<unsigned *>(flag), (unsigned)fail);
    __syncthreads();
    const int f = *flag;
    return f == 0x7fffffff ? -1 : f;
}

// ------------------------------------------------------------------ blocked Cholesky on a packed factor in memory
// Large states (N > 80): the factor does not fit registers or LDS, it lives (packed, lower) in a
// global workspace that stays in L2 / Infinity Cache.  Left-looking, 16 columns per block step:
//   1. panel = A[:, J] - L[:, 0:J] L[J, 0:J]^T   one MFMA chain per row tile, fragments read from the
//      already finished columns in memory, result to an LDS panel ((n - 16J) x 16, ld 17)
//   2. the 16x16 diagonal tile is factored by wave 0 in registers, lane = row (chol_rows)
//   3. the rows below are solved against it, one thread per row, and written out.
template <int NTHREADS, class InitFn>
__device__ __forceinline__ int chol_blocked_mem(double *Lp, int n, double *panel, double *cb, int *flag,
                                                int tid, InitFn init)
{
    constexpr int NW = NTHREADS / 64;
    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int ntc = (n + 15) / 16;
    double *L11 = cb + 4 * 34;            // packed 16x16 factor of the current diagonal tile
    int fail = -1;
    for (int J = 0; J < ntc; ++J) {
        const int c0 = 16 * J, ncol = (n - c0 < 16) ? (n - c0) : 16;
        for (int I = J + wave; I < ntc; I += NW) {
            d4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = 16 * I + g + 4 * r, col = c0 + c;
                double v;
                if (row < n && col < n) v = (row >= col) ? init(row, col) : init(col, row);
                else v = (row == col) ? 1.0 : 0.0;
                acc[r] = v;
            }
            const int ra = 16 * I + c, rb = c0 + c;
            const bool oka = ra < n, okb = rb < n;
            int kk = 0;
            for (; kk + 32 <= c0; kk += 32) {              // eight k-steps per trip: sixteen loads in flight (the factor may
                double af[8], bf[8];                        // live in the global workspace: latency, not bandwidth)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int col = kk + 4 * u + g;
                    const double av = Lp[oka ? pk(n, ra, col) : 0], bv = Lp[okb ? pk(n, rb, col) : 0];
                    af[u] = oka ? -av : 0.0;
                    bf[u] = okb ? bv : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], acc, 0, 0, 0);
            }
            for (; kk < c0; kk += 16) {                    // c0 is a multiple of 16: four k-steps, loads first
                double af[4], bf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int col = kk + 4 * u + g;
                    const double av = Lp[oka ? pk(n, ra, col) : 0], bv = Lp[okb ? pk(n, rb, col) : 0];
                    af[u] = oka ? -av : 0.0;
                    bf[u] = okb ? bv : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) panel[(16 * (I - J) + g + 4 * r) * 17 + c] = acc[r];
        }
        __syncthreads();
        if (wave == 0) {
            // (the panel's first sixteen rows carry a unit diagonal beyond ncol already)
            const int f0 = chol_rows<16>(L11, 16, lane, [&](int i, int j) { return panel[i * 17 + j]; }, ncol);
            if (lane == 0) *flag = f0;
            if (lane < ncol) L11[136 + lane] = 1.0 / L11[pk(ncol, lane, lane)];     // reciprocal pivots for the row solves
        }
        __syncthreads();
        if (*flag >= 0) { fail = c0 + *flag; break; }
        for (int r = tid; r < n - c0; r += NTHREADS) {
            const int row = c0 + r;
            if (r < ncol) {
                for (int b = 0; b <= r; ++b) Lp[pk(n, row, c0 + b)] = L11[pk(ncol, r, b)];
            } else {
                double x[16];
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    x[b] = 0.0;
                    if (b < ncol) {
                        double sum = panel[r * 17 + b];
#pragma unroll
                        for (int q = 0; q < b; ++q) sum -= x[q] * L11[pk(ncol, b, q)];
                        x[b] = sum * L11[136 + b];
                        Lp[pk(n, row, c0 + b)] = x[b];
                    }
                }
            }
        }
        __syncthreads();
    }
    return fail;
}

// ------------------------------------------------------------------ covariance downdate as a factor update
// applyDelta's Cholesky (Msckf.hpp:262-263 -> :659-662) factors Pk - K S K^T.  With Pk = L L^T, covXZ = L A
// (A = 1/2 (Z_{2j+1} - Z_{2j+2})_j, exact while no rotation column wraps) and S = Ls Ls^T:
//     Pk - K S K^T = L (I - B B^T) L^T,   B = A Ls^-T   (N x m', m' <= 8 surviving rows),
// so the new factor is L' = L M with M = chol(I - B B^T): lower triangular, positive diagonal -- the same unique
// factor Eigen::LLT would return for the downdated matrix, up to rounding.  M is identity plus rank m' structure:
//     M_jj = sqrt(d_j),  M_ij = b_i . w_j (i > j),
//     G_j = I - sum_{k<j} b_k b_k^T  (8 x 8),  q_j = G_j^-1 b_j,  d_j = 1 - b_j . q_j,  w_j = -q_j / sqrt(d_j)
// (Schur complements of I - B B^T through the Woodbury identity).  The G_j are PREFIX sums over the rows of B: one
// wave scans them across its lanes (lane j = column j) and every lane factors its own 8 x 8 matrix -- no N-step serial
// chain, no workgroup barrier -- and L M is a triangular matrix product on the matrix cores.  (ldm_prefix / ldm_columns
// below work on T_j = Ls G_j Ls^T, which needs the deviations only.)

// DPP moves of a double (two dwords); invalid source lanes read 0.0 (bound_ctrl)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(h2, l2);
}
// inclusive prefix sum over the 64 lanes of a wave (rows of 16 with row_shr, then row_bcast 15 / 31)
__device__ __forceinline__ double wave_inclusive_scan(double x)
{
    x += dpp_mov_f64<0x111, 0xf>(x);     // row_shr:1
    x += dpp_mov_f64<0x112, 0xf>(x);     // row_shr:2
    x += dpp_mov_f64<0x114, 0xf>(x);     // row_shr:4
    x += dpp_mov_f64<0x118, 0xf>(x);     // row_shr:8
    // (the two cross-row steps move with a full row mask and add under a lane predicate: a partial row mask needs its
    // destination pre-zeroed, and the scheduler hoists those zeros of all 36 scans of ldm_prefix -- 100+ registers)
    const int ln = __lane_id();
    const double t15 = dpp_mov_f64<0x142, 0xf>(x);     // row_bcast:15: last lane of the previous row
    if (ln & 16) x += t15;                               // rows 1 and 3
    const double t31 = dpp_mov_f64<0x143, 0xf>(x);     // row_bcast:31: lane 31
    if (ln & 32) x += t31;                               // rows 2 and 3
    return x;
}

// W buffer: 64 rows x 8 columns, column pairs interleaved so that the two 16-lane halves of an MFMA operand read
// (column 4s + g, g = 0 / 1) fall into one contiguous 32-double span: element (row, c) at (c >> 1) * 128 + 2 * row + (c & 1)
__device__ __forceinline__ int bw_idx(int row, int c) { return (c >> 1) * 128 + 2 * row + (c & 1); }
constexpr int BW_SIZE = 512;

// In terms of the measurement deviations themselves (b_j = Ls^-1 a_j, a_j = 1/2 (Z_{2j+1} - Z_{2j+2})):
//     T_j = S - sum_{k<j} a_k a_k^T,  d_j = 1 - a_j . T_j^-1 a_j,  w~_j = -T_j^-1 a_j / sqrt(d_j),  M_ij = a_i . w~_j,
// so the prefix sums need only the deviations -- they are taken EARLY, by a wave that would otherwise share the S / covXZ
// tiles, and kept in its registers across the gate.  A measurement row the gate rejects drops out by replacing its
// row / column of T_j with the identity and its component of a_j with zero.
// Packed lower triangle of an 8 x 8 matrix: entry (r, c) at r (r + 1) / 2 + c.
#define SLK_G(r, c) Gf[(r) * ((r) + 1) / 2 + (c)]
__device__ __forceinline__ void ldm_prefix(const double (&a)[8], double (&Gf)[36])
{
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            const double e = a[r] * a[c];
            SLK_G(r, c) = wave_inclusive_scan(e) - e;        // exclusive: sum over the lanes (columns) before this one
            if (c == r) __builtin_amdgcn_sched_barrier(0);   // (one row of scans in flight at a time: registers)
        }
}

// One wave, lane j = column j: Gf = prefix sums of ldm_prefix, a = this column's deviations (1/2 dZ), Sm = S (m x m,
// column-major), kept = bit mask of the rows that survived the gate.  Writes Wbuf[j][:] = 1/2 w~_j (the factor 1/2 of
// a_i = 1/2 dZ_i is folded in here, so that the product can read dZ as it stands) and mdiag[j] = sqrt(d_j).
// Returns false if Pk - K S K^T is not positive definite (uniform over the wave).
template <bool FLAT = false>
__device__ __forceinline__ bool ldm_columns(double (&Gf)[36], const double (&a)[8], const double *Sm, int m, unsigned kept,
                                            int lane, int N, double *Wbuf, double *mdiag)
{
    // FLAT: `lane` is the column index of a multi-wave caller (factor_update_blocks), Wbuf is [N][8]
    const bool live = lane < N;
    double b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool kr = (kept >> r) & 1u;
        b[r] = kr ? a[r] : 0.0;
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            const bool in = kr && ((kept >> c) & 1u);
            const double s = Sm[in ? r + m * c : 0];
            SLK_G(r, c) = in ? s - SLK_G(r, c) : ((r == c) ? 1.0 : 0.0);
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
        double s = b[j];
#pragma unroll
        for (int p = 0; p < j; ++p) s = fma(-SLK_G(j, p), y[p], s);
        y[j] = s * rs;
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
        double s = y[c];
#pragma unroll
        for (int p = c + 1; p < 8; ++p) s = fma(-SLK_G(p, c), y[p], s);
        y[c] = s * SLK_G(c, c);
        if (FLAT) { if (live) Wbuf[lane * 8 + c] = y[c] * sc; }
        else Wbuf[bw_idx(lane, c)] = live ? y[c] * sc : 0.0;
    }
    if (live) mdiag[lane] = sqd;
    // (lanes beyond N carry a = 0 and the full prefix: d = 1, they never fail the test unless the real columns do)
    return __all(ok) != 0;
}
#undef SLK_G

// The same factor update for states beyond one wave of columns (N > 64), in O(N^2 m) instead of the O(N^3) factorisation
// of Pk - K S K^T.  With M_jj = sqrt(d_j), M_ij = dZ_i . W_j (i > j) and the columns in blocks of 16:
//     L'(:, J) = L(:, J) M_JJ + U_J W_J^T,   U_J = sum over the column blocks right of J of L(:, J') dZ_J'  (N x 8),
// so a tile row of the factor is a recurrence of its own over its column blocks, right to left, ten MFMAs per tile:
// (L_IJ M_JJ)^T = M_JJ^T L_IJ^T (4), + W_J U^T (2), U^T += dZ_J^T L_IJ^T (4) -- all three take the SAME fragment of the
// factor tile as their B operand, and U^T stays in the accumulator layout the second product reads it in.
// Thread j owns column j for the prefix sums (per-wave scan + wave totals through LDS) and its 8 x 8 Schur matrix
// (ldm_columns); the M_JJ tiles come from 2 MFMAs each.  scratch: Wb [N][8], totals [NTHREADS / 64][36], Mt [ntc][16 x 17].
// Lp may be the global workspace.  Returns false if Pk - K S K^T is not positive definite.
template <int NTHREADS>
__device__ __forceinline__ bool factor_update_blocks(double *Lp, int N, const double *DZ, int m, const double *Sm, unsigned kept,
                                                     double *Wb, double *mdiag, double *totals, double *Mt, int *flag, int tid)
{
    constexpr int NW = NTHREADS / 64;
    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int ntc = (N + 15) >> 4;
    {
        double av[8], Gf[36];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const bool in = tid < N && cc < m;
            const double dz = DZ[in ? tid * m + cc : 0];
            av[cc] = in ? 0.5 * dz : 0.0;
        }
        ldm_prefix(av, Gf);
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) totals[wave * 36 + r * (r + 1) / 2 + cc] = Gf[r * (r + 1) / 2 + cc] + av[r] * av[cc];
        }
        if (tid == 0) *flag = 1;
        __syncthreads();
        for (int w = 0; w < wave; ++w)
#pragma unroll
            for (int e = 0; e < 36; ++e) Gf[e] += totals[w * 36 + e];
        const bool pd = ldm_columns<true>(Gf, av, Sm, m, kept, tid, N, Wb, mdiag);
        if (!pd && lane == 0) *flag = 0;
    }
    __syncthreads();
    if (*flag == 0) return false;
    // M_JJ tiles: element (i, j) at i * 17 + j
    for (int J = wave; J < ntc; J += NW) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int i = 16 * J + c, cc = 4 * ks + g;
            const double av = (i < N && cc < m) ? DZ[i * m + cc] : 0.0;       // A[row i][k c']
            const double bv = (i < N) ? Wb[i * 8 + cc] : 0.0;                 // B[k c'][col j]: lane c = j
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = g + 4 * q, j = c, gi = 16 * J + i;
            const double v = (i > j) ? acc[q] : ((i == j && gi < N) ? mdiag[gi] : 0.0);
            Mt[J * 272 + i * 17 + j] = (gi < N && 16 * J + j < N) ? v : 0.0;
        }
    }
    __syncthreads();
    // tile rows dealt to the waves in a snake over their cost (row I has I + 1 tiles)
    for (int u0 = 0; u0 < ntc; u0 += 2 * NW) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int u = half ? u0 + 2 * NW - 1 - wave : u0 + wave;
            const int I = ntc - 1 - u;
            if (u >= ntc || I < 0) continue;
            const int row = 16 * I + c;
            d4 UT = {0.0, 0.0, 0.0, 0.0};
            double lf[4], ln[4];
            auto fetch = [&](int J, double (&dst)[4]) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int col = 16 * J + 4 * ks + g;
                    const bool ok = J >= 0 && row < N && col <= row;
                    dst[ks] = ok ? Lp[pk(N, ok ? row : 0, ok ? col : 0)] : 0.0;
                }
            };
            fetch(I, lf);
            for (int J = I; J >= 0; --J) {
                fetch(J - 1, ln);                                   // the next tile while this one is worked on
                const double *mt = Mt + J * 272;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)                      // (L_IJ M_JJ)^T: A[row j][k i] = M_JJ(i, j)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(mt[(4 * ks + g) * 17 + c], lf[ks], acc, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {                    // + W_J U^T: A[row j][k c'] = W(16J + j, c')
                    const int j = 16 * J + c;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((j < N) ? Wb[j * 8 + 4 * ks + g] : 0.0, UT[ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {                    // U^T += dZ_J^T L_IJ^T: A[row c'][k i] = dZ(16J + i, c')
                    const int i = 16 * J + 4 * ks + g;
                    UT = __builtin_amdgcn_mfma_f64_16x16x4f64((i < N && c < m) ? DZ[i * m + c] : 0.0, lf[ks], UT, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {                       // acc: row j = g + 4q of the block, column = factor row
                    const int col = 16 * J + g + 4 * q;
                    if (row < N && col <= row) Lp[pk(N, row, col)] = acc[q];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) lf[ks] = ln[ks];
            }
        }
    }
    __syncthreads();
    return true;
}

// output tile (I, J) of the lower triangle -> the wave that computes it; MFMA cost of a tile is 6 (I - J + 1)
template <int NT> __device__ __forceinline__ constexpr int ldm_tile_wave(int I, int J)
{
    if (NT == 4) return (I == 3) ? J : (I == 0 ? 0 : (J == 0 ? I : 3));          // 30 MFMAs per wave
    return (I == 2) ? (J == 0 ? 0 : (J == 1 ? 1 : 3)) : (I == 0 ? 1 : 2);          // NT == 3: 18 / 18 / 18 / 6
}

// position of tile (I, J) among the tiles of its wave, in (I, J) iteration order: a compile-time accumulator index
template <int NT> __device__ __forceinline__ constexpr int ldm_tile_slot(int I, int J)
{
    int n = 0;
    for (int i = 0; i < NT; ++i)
        for (int j = 0; j <= i; ++j) {
            if (i == I && j == J) return n;
            if (ldm_tile_wave<NT>(i, j) == ldm_tile_wave<NT>(I, J)) ++n;
        }
    return n;
}

// L <- L M on the matrix cores, in place in the packed factor.  Called by all NT <= 4 waves of a 256-thread workgroup;
// two workgroup barriers inside.  acc tiles are U(J, I) = L'(I, J)^T: register r of lane (c, g) is L'[16 I + c][16 J + 4 r + g].
template <int NT>
__device__ __forceinline__ void ldm_product(double *Lp, int N, const double *DZ, int m, const double *Wbuf, const double *mdiag,
                                            int lane, int wave)
{
    const int c = lane & 15, g = lane >> 4, tri_g = (g * (g + 1)) >> 1;
    constexpr int MAXT = (NT == 4) ? 4 : 2;          // most tiles a wave owns
    d4 acc[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            if (ldm_tile_wave<NT>(I, J) != wave) continue;
            d4 a = {0.0, 0.0, 0.0, 0.0};
            const double wf0 = Wbuf[bw_idx(16 * J + c, g)], wf1 = Wbuf[bw_idx(16 * J + c, 4 + g)];
#pragma unroll
            for (int K = J; K <= I; ++K) {
                // M(K, J) = dZ_K W_J^T (+ the diagonal fix): result register r of lane (c, g) is M[16 K + g + 4 r][16 J + c]
                d4 mt = {0.0, 0.0, 0.0, 0.0};
                const int arow = 16 * K + c;
                const bool in0 = arow < N && g < m, in1 = arow < N && 4 + g < m;
                const double a0 = DZ[in0 ? arow * m + g : 0], a1 = DZ[in1 ? arow * m + 4 + g : 0];
                mt = __builtin_amdgcn_mfma_f64_16x16x4f64(in0 ? a0 : 0.0, wf0, mt, 0, 0, 0);
                mt = __builtin_amdgcn_mfma_f64_16x16x4f64(in1 ? a1 : 0.0, wf1, mt, 0, 0, 0);
                if (K == J) {
                    const double dg = mdiag[(16 * J + c < N) ? 16 * J + c : 0];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = g + 4 * r;
                        mt[r] = (row > c) ? mt[r] : ((row == c) ? dg : 0.0);
                    }
                }
                // U(J, I) += M(K, J)^T L(I, K)^T: A = the M tile as it stands, B = fragment of L (lane (c, g): L[16 I + c][16 K + 4 s + g])
                double lf[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    // (K < I: the tile lies below the diagonal, every column is <= every row; only the last tile row can
                    // run past N -- both known at compile time in the unrolled loops)
                    const int row = 16 * I + c, col = 16 * K + 4 * s + g;
                    const bool in = (I < NT - 1 || row < N) && (K < I || col <= row);
                    const double v = Lp[in ? packed_colbase(N, 16 * K + 4 * s, g, tri_g) + row : 0];
                    lf[s] = in ? v : 0.0;
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
            if (ldm_tile_wave<NT>(I, J) != wave) continue;
            const d4 a = acc[ldm_tile_slot<NT>(I, J)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + c, col = 16 * J + 4 * r + g;
                if (row < N && col <= row) Lp[packed_colbase(N, 16 * J + 4 * r, g, tri_g) + row] = a[r];
            }
        }
    __syncthreads();
}

// ------------------------------------------------------------------ small reductions
// sum over i in [0, cnt) of term(i), spread over G consecutive lanes (G = 2^k <= 64); every lane of
// the group gets the total
template <int G, class TermFn>
__device__ __forceinline__ double group_sum(int sub, int cnt, TermFn term)
{
    double s = 0.0;
#pragma unroll 4
    for (int i = sub; i < cnt; i += G) s += term(i);
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

// ------------------------------------------------------------------ registered measurement models
// one work item = (sigma point i, feature f); writes that feature's rows of Z[i*m + ...]
__device__ __forceinline__ void measure_item(const KArgs &a, const Lay &L, const double *mp, const double *mu,
                                             const double *Lp, int i, int f, double *Zrow)
{
    const int n = L.N;
    Sig s = sig_of(i);
    if (a.mm == SLK_MM_FEATURE_PROJ) {
        int tp, sp, b;
        pose_of(L, (int)mp[4 * f + 3], tp, sp, b);
        double px = mu[sp] + pert(Lp, n, nullptr, tp, s);
        double py = mu[sp + 1] + pert(Lp, n, nullptr, tp + 1, s);
        double pz = mu[sp + 2] + pert(Lp, n, nullptr, tp + 2, s);
        Quat q = sigma_quat(L, mu, Lp, nullptr, b, s);
        double lx, ly, lz;
        qrot(qconj(q), mp[4 * f] - px, mp[4 * f + 1] - py, mp[4 * f + 2] - pz, lx, ly, lz);
        Zrow[2 * f] = lx / lz;
        Zrow[2 * f + 1] = ly / lz;
    } else if (a.mm == SLK_MM_POSE_POSITION) {
        int tp, sp, b;
        pose_of(L, (int)mp[0], tp, sp, b);
        for (int c = 0; c < 3 && c < a.m; ++c) Zrow[c] = mu[sp + c] + pert(Lp, n, nullptr, tp + c, s);
    } else { // SLK_MM_VO_RELATIVE (Usckf layout): UsckfUnitTest.cpp:62-86, feature triple f
        double dk[3], di[3];
        for (int c = 0; c < 3; ++c) {
            dk[c] = mu[c] + pert(Lp, n, nullptr, c, s);
            di[c] = mu[26 + c] + pert(Lp, n, nullptr, 24 + c, s);
        }
        Quat qk = sigma_quat(L, mu, Lp, nullptr, 0, s), qi = sigma_quat(L, mu, Lp, nullptr, 2, s);
        double rx, ry, rz;
        so3_boxminus(qk, qi, rx, ry, rz);               // delta_state = statek - statek_i
        Quat dq = so3_exp(rx, ry, rz);                  // ... assigned to a WSingleState: set()
        double fx = mu[39 + 3 * f] + pert(Lp, n, nullptr, 36 + 3 * f, s);
        double fy = mu[39 + 3 * f + 1] + pert(Lp, n, nullptr, 36 + 3 * f + 1, s);
        double fz = mu[39 + 3 * f + 2] + pert(Lp, n, nullptr, 36 + 3 * f + 2, s);
        double ox, oy, oz;
        qmat_apply(dq, fx, fy, fz, ox, oy, oz);
        Zrow[3 * f] = ox + (dk[0] - di[0]);
        Zrow[3 * f + 1] = oy + (dk[1] - di[1]);
        Zrow[3 * f + 2] = oz + (dk[2] - di[2]);
    }
}
// SLK_MM_FEATURE_PROJ with every operand loaded up front (clamped addresses, selects afterwards)
__device__ __forceinline__ int feature_count(const Lay &L, const double *mp, int f, int S)
{
    int tp, sp, b;
    pose_of(L, (int)mp[4 * f + 3], tp, sp, b);
    const int c = 2 * (tp + 6) + 1;
    return c < S ? c : S;
}
__device__ __forceinline__ void feature_proj_item(const Lay &L, const double *mp, const double *mu, const double *Lp,
                                                  int i, int f, double *Zrow)
{
    const int n = L.N;
    int tp, sp, b;
    pose_of(L, (int)mp[4 * f + 3], tp, sp, b);
    const int j = (i > 0) ? ((i - 1) >> 1) : 0;
    const double sgn = (i == 0) ? 0.0 : ((i & 1) ? 1.0 : -1.0);
    const int jb = pkcol(n, j);                                       // pk(n, t, j) = jb + t
    double l[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) l[c] = Lp[(j <= tp + c) ? jb + tp + c : 0];
    double x[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) x[c] = mu[sp + c];
    const double fx = mp[4 * f], fy = mp[4 * f + 1], fz = mp[4 * f + 2];
#pragma unroll
    for (int c = 0; c < 6; ++c) l[c] = sgn * ((j <= tp + c) ? l[c] : 0.0);
    const double px = x[0] + l[0], py = x[1] + l[1], pz = x[2] + l[2];
    const Quat q = qmul(Quat{x[3], x[4], x[5], x[6]}, so3_exp(l[3], l[4], l[5]));
    double lx, ly, lz;
    qrot(qconj(q), fx - px, fy - py, fz - pz, lx, ly, lz);
    Zrow[2 * f] = lx / lz;
    Zrow[2 * f + 1] = ly / lz;
}
__host__ __device__ __forceinline__ int measure_features(int mm, int m)
{
    return mm == SLK_MM_FEATURE_PROJ ? m / 2 : (mm == SLK_MM_POSE_POSITION ? 1 : m / 3);
}

// Pose indices of the registered measurement models are caller data: 0 .. k (Msckf: current state, clones) or 0 .. 2
// (Usckf: statek, statek_l, statek_i).  Anything else (negative, NaN, too large) would index mu / the factor out of
// range: the update is skipped and SLK_ST_BAD_INDEX reported.  Uniform over the workgroup (every thread reads the
// same few parameters).
__device__ __forceinline__ bool pose_params_ok(const KArgs &a, const Lay &L, const double *mp)
{
    const double maxc = (L.kind == SLK_MSCKF) ? (double)L.k : 2.0;
    if (a.mm == SLK_MM_FEATURE_PROJ) {
        bool ok = true;
        for (int f = 0; f < a.m / 2; ++f) {
            const double c = mp[4 * f + 3];
            ok = ok && (c >= 0.0) && (c <= maxc);      // false for NaN
        }
        return ok;
    }
    if (a.mm == SLK_MM_POSE_POSITION) {
        const double c = mp[0];
        return (c >= 0.0) && (c <= maxc);
    }
    return true;
}

// ------------------------------------------------------------------ measurement moments
// Z = h(X) over the implicit sigma points of (mu, L), mean_z, innovation, S = 1/2 dZ dZ^T + R and
// covXZ = 1/2 sum (X_i [-] mu)(Z_i - mean_z)^T  (Msckf.hpp:231-239, Usckf.hpp:277-283).
// Lp = packed Cholesky factor.  *flag must be 0 on entry.
struct NoSpare { __device__ __forceinline__ void operator()() const {} };
// `spare`: work for the LAST wave to run beside the S / covXZ tiles of the others (needs spare_ok, one S tile and no
// wrapped rotation column; the wave then takes no tiles); *spare_ran tells whether it did.
template <int NTHREADS, class DiagFn, class SpareFn = NoSpare>
__device__ __forceinline__ void measurement_moments(const KArgs &a, const Lay &L, int bidx, int tid, const double *mu,
                                                    const double *Lp, double *Z, double *DZ, double *Pxz,
                                                    double *Sm, double *zbar, double *innov, int *flag,
                                                    DiagFn pdiag /* diagonal of the covariance Lp factors */,
                                                    double *red = nullptr, int red_cap = 0,
                                                    bool spare_ok = false, SpareFn spare = SpareFn(), bool *spare_ran = nullptr)
{
    const int N = L.N, m = a.m, S = 2 * N + 1, nso3 = L.nso3;
    const double *mp = a.mp ? a.mp + (size_t)bidx * a.mp_stride : nullptr;
    // Z = h(X): Msckf.hpp:231-232
    if (a.mm == SLK_MODEL_EXTERNAL) {
        const double *Ze = a.Zext + (size_t)bidx * S * m;
        for (int e = tid; e < S * m; e += NTHREADS) Z[e] = Ze[e];
    } else {
        int nf = measure_features(a.mm, m);
        if (a.mm == SLK_MM_FEATURE_PROJ) {
            // a feature seen from pose c depends on that pose's 6 tangent rows only; L is lower triangular, so
            // only the sigma points of columns j <= tp + 5 differ from X_0 there: evaluate those items
            // (flattened over the features), the rest of the feature's rows repeat Z_0
            int total = 0;
            for (int g = 0; g < nf; ++g) total += feature_count(L, mp, g, S);
            for (int e = tid; e < total; e += NTHREADS) {
                int f = 0, base = 0, run = 0;
                for (int g = 0; g < nf; ++g) {
                    if (e >= run) { f = g; base = run; }
                    run += feature_count(L, mp, g, S);
                }
                feature_proj_item(L, mp, mu, Lp, e - base, f, Z + (e - base) * m);
            }
            __syncthreads();
            for (int e = tid; e < S * nf; e += NTHREADS) {
                int f = e % nf, i = e / nf;
                if (i >= feature_count(L, mp, f, S)) { Z[i * m + 2 * f] = Z[2 * f]; Z[i * m + 2 * f + 1] = Z[2 * f + 1]; }
            }
        } else {
            for (int e = tid; e < S * nf; e += NTHREADS) {
                int f = e % nf, i = e / nf;
                measure_item(a, L, mp, mu, Lp, i, f, Z + i * m);
            }
        }
    }
    // rotation columns of L longer than pi make log(exp(v)) wrap (MTK log uses atan): flag them.
    // Cheap bound first: |L(rot rows of block b, j)|^2 <= sum over the block's three rows of |L(row, :)|^2
    // = P(t0,t0) + P(t0+1,t0+1) + P(t0+2,t0+2); only if that reaches pi^2 the columns are looked at.
    for (int b = tid; b < nso3; b += NTHREADS) {
        int t0 = so3_toff(L, b);
        if (pdiag(t0) + pdiag(t0 + 1) + pdiag(t0 + 2) >= 9.869604401089358) *flag = 2;
    }
    __syncthreads();
    if (*flag == 2) {
        __syncthreads();
        if (tid == 0) *flag = 0;
        __syncthreads();
        for (int e = tid; e < N * nso3; e += NTHREADS) {
            int j = e % N, b = e / N, t0 = so3_toff(L, b);
            double v0 = Lz(Lp, N, t0, j), v1 = Lz(Lp, N, t0 + 1, j), v2 = Lz(Lp, N, t0 + 2, j);
            if (v0 * v0 + v1 * v1 + v2 * v2 >= 9.869604401089358) *flag = 1;
        }
        __syncthreads();
    }
    SLK_STAMP_NR(4);
    // mean_z (:234), innovation (:236); DZ
    for (int r = tid / 32; r < m; r += NTHREADS / 32) {
        double sum = group_sum<32>(tid & 31, S, [&](int i) { return Z[i * m + r]; });
        if ((tid & 31) == 0) {
            double zb = sum / (double)S;
            zbar[r] = zb;
            innov[r] = a.z[(size_t)bidx * m + r] - zb;
        }
    }
    for (int e = tid; e < N * m; e += NTHREADS) {
        int r = e % m, j = e / m;
        DZ[e] = Z[(2 * j + 1) * m + r] - Z[(2 * j + 2) * m + r];
    }
    __syncthreads();
    SLK_STAMP_NR(5);
    // S = cov(Z) + R (:238), covXZ (:239 -> :635-657).  X_i [-] mu = +-L.col(j) (and 0 for X_0):
    // covXZ = 1/2 L * (Z_{2j+1} - Z_{2j+2})_j ; exact while every rotation column is shorter than pi.
    // Both are small GEMMs and run on the matrix cores: covXZ row tile I = L[16I.., :] * DZ (only the
    // k-steps up to the diagonal, L is lower triangular), S = dZ^T dZ with K = 2N+1.
    const double *R = a.R + (size_t)bidx * a.r_stride;
    const bool wrap = *flag != 0;
    constexpr int NW = NTHREADS / 64;
    const int lane = tid & 63, wave = tid >> 6, fc = lane & 15, fg = lane >> 4;
    const int ntr = (N + 15) / 16, ntm = (m + 15) / 16;
    const bool ksplit = NW > 1 && ntm == 1 && red && NW * m * m <= red_cap;
    const bool spared = spare_ok && ksplit && !wrap && NW > 1;       // uniform over the workgroup
    const int nwe = spared ? NW - 1 : NW;                            // waves that take tiles
    if (spare_ran) *spare_ran = spared;
    if (spared && wave == NW - 1) {
        // (its own branch up to the first barrier below: what the hook leaves in registers must not span the tile code)
        spare();
        for (int e = lane; e < m * m; e += 64) red[wave * m * m + e] = 0.0;        // no partial S tile from this wave
        SLK_STAMP_NR(16);
    } else {
    if (!wrap) {
        for (int e = wave; e < ntr * ntm && wave < nwe; e += nwe) {
            const int I = e % ntr, jt = e / ntr, row = 16 * I + fc, cz = 16 * jt + fc;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            const int kend = (16 * I + 16 < N) ? (16 * I + 16) : N;
            for (int k0 = 0; k0 < kend; k0 += 16) {
                // four k-steps per trip: all eight operand loads are issued (branch-free: clamped
                // address, then select) before the first MFMA, so LDS latency is paid once per trip
                double af[4], bf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int col = k0 + 4 * u + fg;
                    const bool oka = row < N && col <= row && col < kend, okb = col < kend && cz < m;
                    const double av = Lp[oka ? pk(N, row, col) : 0], bv = DZ[okb ? col * m + cz : 0];
                    af[u] = oka ? av : 0.0;
                    bf[u] = okb ? bv : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int orow = 16 * I + fg + 4 * r;
                if (orow < N && cz < m) Pxz[orow + N * cz] = 0.5 * acc[r];
            }
        }
    } else {
        for (int e = tid; e < N * m; e += NTHREADS) {
            int t = e % N, r = e / N;
            double sum = 0.0;
            int blk = -1, comp = 0, s = t2s(L, t, blk, comp), t0 = t - comp;
            for (int j = 0; j <= t; ++j) {
                double w = 1.0;
                if (s < 0) {
                    double v0 = Lz(Lp, N, t0, j), v1 = Lz(Lp, N, t0 + 1, j), v2 = Lz(Lp, N, t0 + 2, j);
                    double th = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
                    if (th >= 3.141592653589793) w = 2.0 * atan(tan(0.5 * th)) / th;
                }
                sum += w * Lp[pk(N, t, j)] * DZ[j * m + r];
            }
            Pxz[e] = 0.5 * sum;
        }
    }
    SLK_STAMP_NR(16);
    if (ksplit) {
        // one S tile (m <= 16): the 2N+1 sigma points are split over the waves (32 per trip), the
        // partial m x m blocks meet in `red` (scratch of NW * m * m doubles)
        const double za = (fc < m) ? zbar[fc] : 0.0;
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
        for (int k0 = 32 * wave; k0 < S && wave < nwe; k0 += 32 * nwe) {
            double af[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = k0 + 4 * u + fg;
                const bool ok = i < S && fc < m;
                const double av = Z[ok ? i * m + fc : 0];
                af[u] = ok ? av - za : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], af[u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u + 1], af[u + 1], acc1, 0, 0, 0);
            }
        }
        acc0 = acc0 + acc1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = fg + 4 * r;
            if (orow < m && fc < m) red[wave * m * m + fc * m + orow] = acc0[r];
        }
        }
        }       // end of the tile waves' branch
        if (ksplit) {
        __syncthreads();
        for (int e = tid; e < m * m; e += NTHREADS) {
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[w * m * m + e];
            Sm[e] = 0.5 * sum + R[e];
        }
        SLK_STAMP_NR(17);
        __syncthreads();
        SLK_STAMP_NR(18);
        return;
    }
    // S: lower tiles (a >= b) of the m x m matrix
    for (int e = wave; e < ntm * (ntm + 1) / 2; e += NW) {      // wave 0 has the shortest covXZ tile
        int ta = 0;
        while ((ta + 1) * (ta + 2) / 2 <= e) ++ta;
        const int tb = e - ta * (ta + 1) / 2;
        const int ra = 16 * ta + fc, rb = 16 * tb + fc;
        const double za = (ra < m) ? zbar[ra] : 0.0, zb = (rb < m) ? zbar[rb] : 0.0;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        d4 accp[4] = {acc, acc, acc, acc};      // four independent accumulation chains
        for (int k0 = 0; k0 < S; k0 += 32) {
            double af[8], bf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = k0 + 4 * u + fg;
                const bool oka = i < S && ra < m, okb = i < S && rb < m;
                const double av = Z[oka ? i * m + ra : 0], bv = Z[okb ? i * m + rb : 0];
                af[u] = oka ? av - za : 0.0;
                bf[u] = okb ? bv - zb : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) accp[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], accp[u & 3], 0, 0, 0);
        }
        acc = (accp[0] + accp[1]) + (accp[2] + accp[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int orow = 16 * ta + fg + 4 * r, ocol = 16 * tb + fc;
            if (orow < m && ocol < m) {
                Sm[orow + m * ocol] = 0.5 * acc[r] + R[orow + m * ocol];
                if (ta != tb) Sm[ocol + m * orow] = 0.5 * acc[r] + R[ocol + m * orow];
            }
        }
    }
    SLK_STAMP_NR(17);
    __syncthreads();
    SLK_STAMP_NR(18);
}

// ------------------------------------------------------------------ 12-DOF predict phase
// Msckf.hpp:102-165 == Usckf.hpp:117-181: sigma points of the current State, process model map,
// manifold mean, new Pk_i = cov + Q.  pin(i, j) = lower triangle of the 12x12 covariance block;
// Lblk (packed, 78) receives its Cholesky factor (Usckf needs it for Fk); x13 = current State mean,
// replaced by the new mean.  Pn (12x12, ld 12) receives the new block, Pxy (WANT_PXY) the matrix M = L^-1 Pxy.  Returns 0 or status bits
// (uniform), -1 after a sigma-point emission.  Runs in ONE wave (tid = lane < 64), no workgroup barrier.
// scratch (doubles): Ys[25*13] dbuf[25*12] refs[16]  (callers still reserve the 88 doubles behind them)
template <bool WANT_PXY, class PinFn>
__device__ __forceinline__ int predict_phase(const KArgs &a, int bidx, int tid, PinFn pin, double *Lblk, double *x13,
                                             double *Pn, double *scr, double *Pxy /* 12x12 ld 12, only if WANT_PXY */)
{
    double *Ys = scr, *dbuf = scr + 25 * 13, *refs = dbuf + 25 * 12;
    const int fail = chol_rows<12>(Lblk, 12, tid, pin);
    SLK_STAMP_NR(21);
    if (fail >= 0) return SLK_ST_LLT_FAIL;
    const double *u = a.u ? a.u + (size_t)bidx * a.u_stride : nullptr;
    if (tid < 25) {
        Sig s = sig_of(tid);
        double v[12], x[13];                  // (every loop over them unrolled: registers, no scratch memory)
#pragma unroll
        for (int t = 0; t < 12; ++t) v[t] = pert(Lblk, 12, nullptr, t, s);
        state_boxplus(x13, v, x);
        double *yo = Ys + tid * 13;           // f(X_i) goes straight to its LDS row
        if (a.emit == 1) {
#pragma unroll
            for (int c = 0; c < 13; ++c) a.Xout[((size_t)bidx * 25 + tid) * 13 + c] = x[c];
        } else if (a.pm == SLK_MODEL_EXTERNAL) {
#pragma unroll
            for (int c = 0; c < 13; ++c) yo[c] = a.Yext[((size_t)bidx * 25 + tid) * 13 + c];
        } else {
            process_model(a.pm, u, x, yo);
        }
    }
    if (a.emit == 1) return -1;   // sigma points emitted, nothing else to do
    wave_sync();
    // Manifold mean (Msckf.hpp:473-487).  Sigma point tid keeps its quaternion and its vector-row deviations in
    // registers, EVERY lane keeps the reference quaternion and moves it itself (the vector rows of the reference sit in
    // LDS, moved by the lane of their component): per pass one LDS transposition (the 25 x 12 deviations, summed by four
    // lanes per component) instead of three round trips, one logarithm and one exponential.  The vector rows are kept as
    // deviations, d <- d - mean(d) (= a - (ref + mean) up to rounding); the loop test is on the squared norm.
    int it = 0, status = 0;
    Quat q, rq;
    double dv[9], dr[3], mr[3];
    {
        const double *yrow = Ys + (tid < 25 ? tid : 0) * 13;
        q = ldq(yrow + 3);
        rq = ldq(Ys + 3);                                                        // reference = X[0]  (Msckf.hpp:473)
#pragma unroll
        for (int c = 0; c < 9; ++c) { const int sidx = c < 3 ? c : c + 4; dv[c] = yrow[sidx] - Ys[sidx]; }
        if (tid < 13) refs[tid] = Ys[tid];
    }
    SLK_STAMP_NR(22);
    double n2;
    do {                                            // Msckf.hpp:478-487
        so3_boxminus(q, rq, dr[0], dr[1], dr[2]);
        if (tid < 25) {
            double *row = dbuf + tid * 12;
#pragma unroll
            for (int c = 0; c < 3; ++c) { row[c] = dv[c]; row[3 + c] = dr[c]; }
#pragma unroll
            for (int c = 3; c < 9; ++c) row[3 + c] = dv[c];
        }
        wave_sync();
        double mv[9];
        {
            // mean over the 25 points: lane = part * 16 + component, four partial sums of 7 / 6 / 6 / 6 points, two shuffles
            const int comp = tid & 15, part = tid >> 4;
            double sum = 0.0;
            if (comp < 12)
                for (int i = part; i < 25; i += 4) sum += dbuf[i * 12 + comp];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const double mc = sum * (1.0 / 25.0);
#pragma unroll
            for (int c = 0; c < 3; ++c) mr[c] = readlane_f64(mc, 3 + c);
#pragma unroll
            for (int c = 0; c < 9; ++c) mv[c] = readlane_f64(mc, c < 3 ? c : c + 3);
            if (tid < 12 && (tid < 3 || tid >= 6)) refs[tid < 3 ? tid : tid + 1] += mc;
        }
        n2 = ((mv[0] * mv[0] + mv[1] * mv[1]) + (mv[2] * mv[2] + mr[0] * mr[0]))
           + ((mr[1] * mr[1] + mr[2] * mr[2]) + (mv[3] * mv[3] + mv[4] * mv[4]))
           + ((mv[5] * mv[5] + mv[6] * mv[6]) + (mv[7] * mv[7] + mv[8] * mv[8]));
        rq = qmul(rq, so3_exp(mr[0], mr[1], mr[2]));
#pragma unroll
        for (int c = 0; c < 9; ++c) dv[c] -= mv[c];
        wave_sync();                                // (the next pass rewrites dbuf)
    } while (n2 > 1e-12 && ++it < 10000);
    SLK_STAMP_NR(23);
    SLK_NOTE(25, it + 1);
    if (it >= 10000) status |= SLK_ST_MEAN_NOT_CONVERGED;
    // covariance (Msckf.hpp:554-570) + Q (:162).  The loop leaves with |mean_delta| <= 1e-6: the deviations against the FINAL
    // mean follow from the ones just taken -- the vector rows are there already, the rotation block by the first-order
    // correction d' = d - Jl^-1(d) m of the update kernels (error O(|m|^2) <= 1e-12) -- instead of another pass of logarithms.
    if (tid < 25) {
        if (it < 10000) {
            const double m0 = mr[0], m1 = mr[1], m2 = mr[2];
            const double x = dr[0], y = dr[1], z = dr[2];
            const double cx = y * m2 - z * m1, cy = z * m0 - x * m2, cz = x * m1 - y * m0;      // d x m
            const double ax = y * cz - z * cy, ay = z * cx - x * cz, az = x * cy - y * cx;      // d x (d x m)
            const double a12 = 1.0 / 12.0 + (x * x + y * y + z * z) * (1.0 / 720.0);
            dr[0] = x - m0 + 0.5 * cx - a12 * ax;
            dr[1] = y - m1 + 0.5 * cy - a12 * ay;
            dr[2] = z - m2 + 0.5 * cz - a12 * az;
        } else {
            so3_boxminus(q, rq, dr[0], dr[1], dr[2]);
        }
        double *row = dbuf + tid * 12;
#pragma unroll
        for (int c = 0; c < 3; ++c) { row[c] = dv[c]; row[3 + c] = dr[c]; }
#pragma unroll
        for (int c = 3; c < 9; ++c) row[3 + c] = dv[c];
    }
    wave_sync();
    const double *Q = a.Q + (size_t)bidx * a.q_stride;
    {   // 1/2 D D^T, D = 12 x 25, on the matrix cores: one 16x16 tile, seven k-steps of four sigma points
        const int fc = tid & 15, fg = tid >> 4;
        double fr[7];
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int i = 4 * ks + fg;
            const bool ok = fc < 12 && i < 25;
            const double v = dbuf[ok ? i * 12 + fc : 0];
            fr[ks] = ok ? v : 0.0;
        }
        d4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0;
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            if (ks & 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[ks], fr[ks], c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[ks], fr[ks], c0, 0, 0, 0);
        }
        c0 = c0 + c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = fg + 4 * r;
            if (row < 12 && fc < 12) Pn[row + 12 * fc] = 0.5 * c0[r] + Q[row + 12 * fc];
        }
    }
    for (int e = tid; WANT_PXY && e < 144; e += 64) {
        // Pxy = 1/2 sum (XCopy_i [-] mu_old)(X_i [-] mu_new)^T with XCopy_i [-] mu_old = +-L.col(j) (Usckf.hpp:152-153,
        // :691-712), i.e. Pxy = L M with M(j, c) = 1/2 (d_{+j}(c) - d_{-j}(c)): the callers want Fk^T = Pk_i^-1 Pxy =
        // L^-T M (:154) -- M itself is handed out (element (j, c) at j + 12 c) and the forward substitution never runs
        const int j = e % 12, c = e / 12;
        Pxy[e] = 0.5 * (dbuf[(2 * j + 1) * 12 + c] - dbuf[(2 * j + 2) * 12 + c]);
    }
    wave_sync();
    if (tid < 13 && (tid < 3 || tid >= 7)) x13[tid] = refs[tid];
    if (tid == 0) stq(x13 + 3, rq);
    wave_sync();
    return status;
}

// ------------------------------------------------------------------ fp64 MFMA covariance rebuild
// P+ = 1/2 * D * D^T (Msckf.hpp:574-589), D = N x S deviations streamed through LDS in panels of KP
// sigma points.  v_mfma_f64_16x16x4_f64 (64 cycles per SIMD, measured): lane l holds A[row l&15][k l>>4]
// and B[k l>>4][col l&15]; for D*D^T the A fragment of row-tile I equals the B fragment of
// column-tile I.  Result lane l, register r: row (l>>4)+4r, col l&15.  Lower-triangle tiles only,
// dealt round-robin to the waves.
template <int NT> struct TileMap {
    static constexpr int NTILES = NT * (NT + 1) / 2;
    __host__ __device__ static constexpr int row(int t) { int i = 0; while ((i + 1) * (i + 2) / 2 <= t) ++i; return i; }
    __host__ __device__ static constexpr int col(int t) { return t - row(t) * (row(t) + 1) / 2; }
};

// tiles are dealt round-robin to the waves; with more than TPWMAX tiles per wave the rebuild runs in
// passes of NW * TPWMAX tiles (large states)
template <int NT, int NW> struct TilePlan {
    static constexpr int TPWMAX = (NT > 6) ? 12 : 6;      // the large-state kernel runs one workgroup per CU: registers to spare
    static constexpr int TPW_ALL = (TileMap<NT>::NTILES + NW - 1) / NW;
    static constexpr int TPW = TPW_ALL < TPWMAX ? TPW_ALL : TPWMAX;
    static constexpr int PER_PASS = NW * TPW;
    static constexpr int PASSES = (TileMap<NT>::NTILES + PER_PASS - 1) / PER_PASS;
};

template <int NT, int NW, int T>
struct MfmaTiles {
    static constexpr int TPW = TilePlan<NT, NW>::TPW;
    static constexpr int PER_PASS = TilePlan<NT, NW>::PER_PASS;
    __device__ __forceinline__ static void run(const double (&frag)[NT], d4 (&acc)[TPW], int wave, int pass)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave && (T / PER_PASS) == pass)
                acc[(T % PER_PASS) / NW] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[TileMap<NT>::row(T)], frag[TileMap<NT>::col(T)],
                                                                              acc[(T % PER_PASS) / NW], 0, 0, 0);
            MfmaTiles<NT, NW, T + 1>::run(frag, acc, wave, pass);
        }
    }
    // accumulators -> global P (column-major, both triangles).  The tile is written transposed
    // (P is symmetric), so that the 16 lanes of a row group store 128 contiguous bytes.
    __device__ __forceinline__ static void store(double *gP, int N, const d4 (&acc)[TPW], int wave, int lane, int pass,
                                                 double scale = 0.5)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave && (T / PER_PASS) == pass) {
                constexpr int I = TileMap<NT>::row(T), J = TileMap<NT>::col(T);
                int c = 16 * J + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int rr = 16 * I + (lane >> 4) + 4 * r;
                    if (rr < N && c < N) {
                        double v = scale * acc[(T % PER_PASS) / NW][r];
                        gP[c + (size_t)rr * N] = v;
                        if (I != J) gP[rr + (size_t)c * N] = v;
                    }
                }
            }
            MfmaTiles<NT, NW, T + 1>::store(gP, N, acc, wave, lane, pass, scale);
        }
    }
    // accumulator tiles -> LDS, tile T at T * 272, element (row, col) at col * 17 + row
    __device__ __forceinline__ static void to_lds(double *dst, const d4 (&acc)[TPW], int wave, int lane)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[T * 272 + (lane & 15) * 17 + (lane >> 4) + 4 * r] = acc[T / NW][r];
            }
            MfmaTiles<NT, NW, T + 1>::to_lds(dst, acc, wave, lane);
        }
    }
    // rotation-row index of a tangent row of the Msckf layout (-1: a vector row): State.hpp:141-149, :246-252, :384-396
    __device__ __forceinline__ static int rho_of(int t)
    {
        if (t < 12) return (t >= 3 && t < 6) ? t - 3 : -1;
        const int cc = (t - 12) / 6, r = (t - 12) - 6 * cc;
        return r >= 3 ? 3 * cc + r : -1;
    }
    // store (scale 1) plus the matrix EEt given in the index space of the rotation rows (lower tiles as written by to_lds)
    __device__ __forceinline__ static void store_plus_rot(double *gP, int N, const d4 (&acc)[TPW], int wave, int lane, const double *EEt)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave) {
                constexpr int I = TileMap<NT>::row(T), J = TileMap<NT>::col(T);
                const int c = 16 * J + (lane & 15), rc = rho_of(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * I + (lane >> 4) + 4 * r;
                    if (rr < N && c < N) {
                        const int r1 = rho_of(rr);
                        double v = acc[T / NW][r];
                        if (r1 >= 0 && rc >= 0) {
                            const int hi = r1 > rc ? r1 : rc, lo = r1 > rc ? rc : r1;
                            const int ti = hi >> 4, tj = lo >> 4;
                            v += EEt[(ti * (ti + 1) / 2 + tj) * 272 + (lo & 15) * 17 + (hi & 15)];
                        }
                        gP[c + (size_t)rr * N] = v;
                        if (I != J) gP[rr + (size_t)c * N] = v;
                    }
                }
            }
            MfmaTiles<NT, NW, T + 1>::store_plus_rot(gP, N, acc, wave, lane, EEt);
        }
    }
};

// Reduced-precision variants of the rebuild for the precision sweep of BASELINE config 5 (never the
// parity path): fp32 operands on v_mfma_f32_16x16x4_f32, or bf16 operands / fp32 accumulation on
// v_mfma_f32_16x16x32_bf16.  fp32 result layout: lane l, register r -> row 4*(l>>4) + r, col l&15.
template <int NT, int NW, int T>
struct MfmaTiles32 {
    static constexpr int TPW = TilePlan<NT, NW>::TPW;
    static constexpr int PER_PASS = TilePlan<NT, NW>::PER_PASS;
    __device__ __forceinline__ static void run(const float (&frag)[NT], f4 (&acc)[TPW], int wave, int pass)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave && (T / PER_PASS) == pass)
                acc[(T % PER_PASS) / NW] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[TileMap<NT>::row(T)], frag[TileMap<NT>::col(T)],
                                                                             acc[(T % PER_PASS) / NW], 0, 0, 0);
            MfmaTiles32<NT, NW, T + 1>::run(frag, acc, wave, pass);
        }
    }
    __device__ __forceinline__ static void run_bf16(const b8 (&frag)[NT], f4 (&acc)[TPW], int wave, int pass)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave && (T / PER_PASS) == pass)
                acc[(T % PER_PASS) / NW] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag[TileMap<NT>::row(T)], frag[TileMap<NT>::col(T)],
                                                                               acc[(T % PER_PASS) / NW], 0, 0, 0);
            MfmaTiles32<NT, NW, T + 1>::run_bf16(frag, acc, wave, pass);
        }
    }
    // accumulator tiles -> LDS (doubles), tile T at T * 272, element (row, col) at col * 17 + row
    __device__ __forceinline__ static void to_lds(double *dst, const f4 (&acc)[TPW], int wave, int lane)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[T * 272 + (lane & 15) * 17 + 4 * (lane >> 4) + r] = (double)acc[T / NW][r];
            }
            MfmaTiles32<NT, NW, T + 1>::to_lds(dst, acc, wave, lane);
        }
    }
    // store (scale 1) plus the matrix EEt in the index space of the rotation rows (see MfmaTiles::store_plus_rot)
    __device__ __forceinline__ static void store_plus_rot(double *gP, int N, const f4 (&acc)[TPW], int wave, int lane, const double *EEt)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave) {
                constexpr int I = TileMap<NT>::row(T), J = TileMap<NT>::col(T);
                const int c = 16 * J + (lane & 15), rc = MfmaTiles<NT, NW, 0>::rho_of(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * I + 4 * (lane >> 4) + r;
                    if (rr < N && c < N) {
                        const int r1 = MfmaTiles<NT, NW, 0>::rho_of(rr);
                        double v = (double)acc[T / NW][r];
                        if (r1 >= 0 && rc >= 0) {
                            const int hi = r1 > rc ? r1 : rc, lo = r1 > rc ? rc : r1;
                            const int ti = hi >> 4, tj = lo >> 4;
                            v += EEt[(ti * (ti + 1) / 2 + tj) * 272 + (lo & 15) * 17 + (hi & 15)];
                        }
                        gP[c + (size_t)rr * N] = v;
                        if (I != J) gP[rr + (size_t)c * N] = v;
                    }
                }
            }
            MfmaTiles32<NT, NW, T + 1>::store_plus_rot(gP, N, acc, wave, lane, EEt);
        }
    }
    __device__ __forceinline__ static void store(double *gP, int N, const f4 (&acc)[TPW], int wave, int lane, int pass)
    {
        if constexpr (T < TileMap<NT>::NTILES) {
            if ((T % NW) == wave && (T / PER_PASS) == pass) {
                constexpr int I = TileMap<NT>::row(T), J = TileMap<NT>::col(T);
                int c = 16 * J + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int rr = 16 * I + 4 * (lane >> 4) + r;
                    if (rr < N && c < N) {
                        double v = 0.5 * (double)acc[(T % PER_PASS) / NW][r];
                        gP[c + (size_t)rr * N] = v;
                        if (I != J) gP[rr + (size_t)c * N] = v;
                    }
                }
            }
            MfmaTiles32<NT, NW, T + 1>::store(gP, N, acc, wave, lane, pass);
        }
    }
};

// ------------------------------------------------------------------ the Msckf step kernel
// predict (optional) + UKF update with applyDelta (optional), one workgroup per filter.
// KST >= 0: the number of sensor-pose clones is a compile-time constant (exact-shape instantiation: the layout and
// packed-index arithmetic fold); KST < 0: taken from the arguments.
// MST > 0: the number of measurement rows is a compile-time constant too (every LDS offset of the carve folds).
// exact-shape fast path of the update + applyDelta (slk_step_fast.hpp, included at the end of this file): k = 4 .. 8 clones,
// m = 8 rows; returns false -- before its first global write -- whenever the general body below has to run instead
template <int K> __device__ __forceinline__ bool msckf_step_fast(const KArgs &a, double *smem);
template <int NT, int NTHREADS, int KST, int MST> struct HasFastStep {
    static constexpr bool value = KST >= 4 && KST <= 8 && MST == 8 && NTHREADS == 256 && NT == (12 + 6 * KST + 15) / 16;
};

template <int NT, int NTHREADS, int KST = -1, int MST = 0>
// (register budget: the exact-shape instantiations (k and m known) fit 128 registers = four workgroups per CU; the run-time
// shapes of N = 33 .. 64 need more live index arithmetic and get 256 = two workgroups per CU instead of scratch memory)
__global__ __launch_bounds__(NTHREADS, (NTHREADS >= 256 && NT >= 3 && NT <= 4 ? (MST > 0 ? SLK_WGS : 2) : ((NT >= 5 && NT <= 8) ? 2 : (NT <= 2 ? 4 : 1)))) void msckf_step_kernel(KArgs a)
{
    constexpr bool BIG = NT > 4;                           // large state (N > 64): factor + rotation store in the global workspace
    extern __shared__ __attribute__((aligned(16))) double smem[];
#ifndef SLK_NO_FAST_STEP
    if constexpr (HasFastStep<NT, NTHREADS, KST, MST>::value) {
        if (!a.do_predict && a.do_update && msckf_step_fast<KST>(a, smem)) return;
    }
#endif
    constexpr int NW = NTHREADS / 64;
    constexpr int GD = Grid<NTHREADS>::GD;
    constexpr int SDN = (16 * NT + GD - 1) / GD;          // Cholesky register slots per dimension
    constexpr int SDM = (MAXM + GD - 1) / GD;
    constexpr int NROWS = (NT <= 2 && KST >= 0) ? 12 + 6 * KST : 1;      // rows of the one-wave register Cholesky (exact shapes, N <= 32)
    static_assert(NT > 2 || NTHREADS == 64, "states of N <= 32 run one wave per filter");
    const int bidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Lay L = a.lay;
    L.kind = SLK_MSCKF;                                    // this kernel is the Msckf step: fold the layout branches
    if constexpr (KST >= 0) { L.k = KST; L.N = 12 + 6 * KST; L.Nq = 13 + 7 * KST; L.nso3 = 1 + KST; }
    if constexpr (MST > 0) { a.m = MST; a.emit = 0; a.do_update = 1; if (NT <= 4) a.rebuild_prec = 0; }     // guaranteed by the launcher (large states: any rebuild precision)
    const int N = L.N, Nq = L.Nq, m = a.m, nso3 = L.nso3;
    const Carve cv = carve_step(L, m, NT, BIG, (KST >= 0 && MST > 0 && !BIG) ? 0 : a.rebuild_prec);
    const int S = cv.S, LDD = cv.LDD;
    double *Lp = BIG ? a.wsL + (size_t)bidx * pk_size(N) : smem + cv.Lp;
    double *mu = smem + cv.mu, *ref = smem + cv.ref;
    double *delta = smem + cv.delta, *md = smem + cv.md, *colbuf = smem + cv.colbuf, *pool = smem + cv.pool, *cq = smem + cv.cq;
    int *ish = reinterpret_cast<int *>(smem + cv.small);      // [0..MAXM) idx, [40] count, [41] outliers, [42] flag, [43] predicted
    int *roff = ish + 48;                                     // nso3 + 1 prefix offsets of the rotation items
    double *gmean = a.mean + (size_t)bidx * Nq;
    double *gP = a.P + (size_t)bidx * N * N;
    double *omean = a.mean_out ? a.mean_out + (size_t)bidx * Nq : gmean;     // where applyDelta's results go
    double *oP = a.P_out ? a.P_out + (size_t)bidx * N * N : gP;
    int status = 0;
    if (a.do_update && a.emit != 4 && tid == 0) a.outliers[bidx] = 0u;
    SLK_STAMP(0);

    for (int e = tid; e < Nq; e += NTHREADS) mu[e] = gmean[e];
    if (tid == 0) {
        int o = 0;
        for (int b = 0; b < nso3; ++b) { roff[b] = o; o += rot_count(L, b); }
        roff[nso3] = o;
        ish[42] = 0;
    }
    __syncthreads();
    SLK_STAMP(1);

    // The predict step (Msckf.hpp:89-189) runs in its own launch (msckf_predict_kernel below): it touches the current
    // state's 12 x 12 block only, keeps one wave busy per filter and nothing of it is reused on chip -- beside the
    // four-wave phases of this kernel it held three waves and 38 KB of LDS idle for 13 % of the step.
    constexpr bool WCHOL = NT >= 3 && NT <= 4 && NW >= NT;       // one tile row per wave
    // N <= 32: the workgroup IS one wave, the predict chain runs here (one launch per step: the step time of these shapes
    // is one filter's latency); the update reads the predicted block from LDS
    bool pred12 = false;
    const double *Pn12 = smem + cv.pn12;
    if constexpr (NT <= 2) {
        if (a.do_predict) {
            double *Lblk = pool, *Pn = pool + 160, *scr = pool + 320;          // 320 + 736 <= PRED_SCRATCH
            const int st = predict_phase<false>(a, bidx, tid, [&](int i, int j) { return gP[i + (size_t)j * N]; }, Lblk, mu, Pn, scr,
                                                nullptr);
            if (st < 0) return;                              // sigma points emitted
            status |= st;
            if (!(st & SLK_ST_LLT_FAIL)) {                   // else: predict skipped, filter unchanged
                for (int e = tid; e < 144; e += NTHREADS) {
                    smem[cv.pn12 + e] = Pn[e];
                    gP[(e % 12) + (size_t)(e / 12) * N] = Pn[e];
                }
                if (tid < 13) gmean[tid] = mu[tid];
                pred12 = true;
            }
            __syncthreads();
        }
    }
    // lower-triangle element of the covariance (only the lower triangle of Pk is ever read: LLT at :412, :447)
    // (both sources are fetched and the VALUE is selected: a select between an LDS and a global pointer turns every access
    // into a flat load with a 64-bit address of its own, kept alive from the first factorisation to the second)
    auto Pin = [&](int i, int j) -> double {
        const double g = gP[i + (size_t)j * N];
        if constexpr (NT <= 2) {
            const double l = Pn12[min(i, 11) + 12 * min(j, 11)];
            return (pred12 && i < 12 && j < 12) ? l : g;
        }
        return g;
    };
    // exact shapes of N <= 32: the one-wave register Cholesky of Pk - (downdate), lane = row (chol_rows hands row
    // min(lane, N - 1) to its element function; down(j) is that row's downdate of column j).  After a predict in this
    // launch the 12 x 12 block comes from LDS -- for N = 12 nothing is fetched from global memory at all.
    auto chol_state = [&](auto down) -> int {
        if (pred12)
            return chol_rows<NROWS>(Lp, NROWS, lane, [&](int i, int j) -> double {
                if constexpr (NROWS == 12) return Pn12[i + 12 * j] - down(j);
                if (j < 12) {
                    const double l = Pn12[min(i, 11) + 12 * j], g = gP[i + (size_t)j * N];
                    return (i < 12 ? l : g) - down(j);
                }
                return gP[i + (size_t)j * N] - down(j);
            });
        return chol_rows<NROWS>(Lp, NROWS, lane, [&](int i, int j) -> double { return gP[i + (size_t)j * N] - down(j); });
    };

    SLK_STAMP(2);
    if (a.do_update || a.emit >= 2) {
        // ---- sigma points of the full state: Msckf.hpp:228-229 -> :400-431
        int fail;
#ifndef SLK_MSCKF_FACTOR_KERNEL
        constexpr bool FACTOR_INSIDE = true;
#else
        constexpr bool FACTOR_INSIDE = false;
#endif
        if (FACTOR_INSIDE && WCHOL && !a.wsfail) {
            // no factor in the workspace (the exact-shape launch factors inside the fast path, and this body is its fallback):
            // one wave, panel by rows, straight into LDS
            if constexpr (WCHOL && FACTOR_INSIDE) {
                if (wave == 0) {
                    d4 acc[CholM<NT>::NTL];
                    cholm_load_t<NT>(acc, N, lane, Pin);
                    const int f0 = cholp_factor<NT, 0>(acc, Lp, N, colbuf, lane);
                    if (lane == 0) ish[45] = f0;
                }
            }
            __syncthreads();
            fail = ish[45];
        } else if constexpr (WCHOL) {
            // the factor comes from msckf_chol_kernel (its own launch, one wave per filter at twelve filters per CU),
            // packed, through a workspace
            const double *gL = a.wsL + (size_t)bidx * pk_size(N);
            for (int e0 = 0; e0 < pk_size(N); e0 += 8 * NTHREADS) {         // eight loads in flight per thread
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int e = e0 + q * NTHREADS + tid; v[q] = (e < pk_size(N)) ? gL[e] : 0.0; }
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int e = e0 + q * NTHREADS + tid; if (e < pk_size(N)) Lp[e] = v[q]; }
            }
            fail = a.wsfail[bidx];
            __syncthreads();
        } else if constexpr (NT <= 2 && KST >= 0) {
            fail = chol_state([](int) { return 0.0; });                // (one wave per filter: no barrier)
        } else if constexpr (NT <= 4) {
            if (wave == 0) {
                d4 acc[CholM<NT>::NTL];
                cholm_load<NT>(acc, N, lane, Pin);
                int f0 = cholm_factor<NT>(acc, Lp, N, colbuf, lane);
                if (lane == 0) ish[45] = f0;
            }
            __syncthreads();
            fail = ish[45];
        } else if constexpr (BIG) {
            // (the launcher ran msckf_chol_big_kernel: the factor is in the workspace already)
            if (a.wsfail) fail = a.wsfail[bidx];
            else fail = chol_blocked_mem<NTHREADS>(Lp, N, pool, colbuf, &ish[45], tid, Pin);
        } else {
            fail = chol_packed<NTHREADS, SDN>(Lp, N, colbuf, tid, Pin);
        }
        SLK_STAMP(3);
        bool redraw = false;
        if (fail >= 0) {
            status |= SLK_ST_LLT_FAIL;
        } else if (a.emit == 3) {
            // checkSigmaPoints (Msckf.hpp:819-839): re-draw (mu, 0, Pk), manifold mean and covariance go to the
            // caller's scratch (a.mean_out / a.P_out), the filter itself is left as it is
            for (int t = tid; t < N; t += NTHREADS) delta[t] = 0.0;
            __syncthreads();
            redraw = true;
        } else if (a.emit == 2) {
            double *X = a.Xout + (size_t)bidx * S * Nq;
            for (int e = tid; e < S * N; e += NTHREADS) {
                int t = e % N, i = e / N, blk = 0, comp = 0;
                int s = t2s(L, t, blk, comp);
                if (s >= 0) X[(size_t)i * Nq + s] = mu[s] + pert(Lp, N, nullptr, t, sig_of(i));
            }
            for (int e = tid; e < S * nso3; e += NTHREADS) {
                int b = e % nso3, i = e / nso3;
                Quat q = sigma_quat(L, mu, Lp, nullptr, b, sig_of(i));
                double *o = X + (size_t)i * Nq + so3_soff(L, b);
                o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
            }
        } else if (!pose_params_ok(a, L, a.mp ? a.mp + (size_t)bidx * a.mp_stride : nullptr)) {
            status |= SLK_ST_BAD_INDEX;                       // pose index of a registered model out of range: update skipped
        } else {
            // ---- pool carve for the measurement part
            double *Z = pool;                                   // [S][m]
            double *DZ = Z + z_region(S, m, NT, BIG);           // [N][m]: Z_{2j+1} - Z_{2j+2}
            double *Pxz = DZ + round_up(N * m, 2);              // N x m (ld N)
            double *K = Z;                                      // N x m' (ld N): written after the moments, Z is dead then
            double *Sm = Pxz + round_up(N * m, 2);              // m x m (ld m)
            double *G = Sm + round_up(m * m, 2);                // packed factor of S / Gauss-Jordan tableau
            double *zbar = G + round_up(m * (2 * m + 1), 2);
            double *innov = zbar + round_up(m, 2);
            int *idx = ish;
            // factor-update path (see ldm_columns): the last wave scans the prefix sums of the measurement deviations
            // while the others do the S / covXZ tiles, and keeps them in registers across the gate
            // (run-time shapes: the scan runs after the gate, right before its columns -- its 88 registers across the gate
            // on top of their index arithmetic would not fit)
            constexpr bool LATE_SCAN = MST == 0;
            double ldm_a[8], ldm_pf[36];
            bool ldm_on = false;
            auto ldm_scan = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool in = lane < N && c < m;
                    const double dz = DZ[in ? lane * m + c : 0];
                    ldm_a[c] = in ? 0.5 * dz : 0.0;
                }
                ldm_prefix(ldm_a, ldm_pf);
            };
            // A measurement that is LINEAR in the tangent space (the position fix of a pose: Z_i = mu_pos +- L(tp + r, j)) has its
            // sigma-point moments in closed form -- mean_z = h(mu), S = (L L^T)(rows, rows) + R = P(rows, rows) + R, covXZ =
            // L L(rows, :)^T = P(:, rows) (Msckf.hpp:231-239 in exact arithmetic; rounding apart, 1e-16) -- as long as no rotation
            // column can wrap (X_i [-] mu = +-L(:, j)): the exact-shape kernel of BASELINE config 2 (N = 12, three rows) takes it
            bool closed_form = false;
            if constexpr (NT == 1 && KST == 0 && MST == 3) {
                if (a.mm == SLK_MM_POSE_POSITION && a.emit == 0) {
                    // (after a predict in this launch the whole matrix is the 12 x 12 block in LDS: no global access)
                    auto closed = [&](auto Pacc) __attribute__((always_inline)) -> bool {
                        bool nowrap = true;
                        for (int b = 0; b < nso3; ++b) {
                            const int t0 = so3_toff(L, b);
                            nowrap = nowrap && (Pacc(t0, t0) + Pacc(t0 + 1, t0 + 1) + Pacc(t0 + 2, t0 + 2) < 9.869604401089358);
                        }
                        if (!nowrap) return false;                      // (uniform)
                        SLK_STAMP_NR(4);
                        SLK_STAMP_NR(5);
                        const double *mp = a.mp + (size_t)bidx * a.mp_stride;
                        const double *R = a.R + (size_t)bidx * a.r_stride;
                        int tp, sp, pb;
                        pose_of(L, (int)mp[0], tp, sp, pb);
                        if (tid < 3) {
                            const double zb = mu[sp + tid];
                            zbar[tid] = zb;
                            innov[tid] = a.z[(size_t)bidx * 3 + tid] - zb;
                        }
                        if (tid < 9) {
                            const int r = tid % 3, c = tid / 3, i = tp + (r > c ? r : c), j = tp + (r > c ? c : r);
                            Sm[r + 3 * c] = Pacc(i, j) + R[r + 3 * c];
                        }
                        if (tid < 3 * N) {
                            const int t = tid % N, c = tid / N, i = t > tp + c ? t : tp + c, j = t > tp + c ? tp + c : t;
                            Pxz[t + N * c] = Pacc(i, j);
                        }
                        __syncthreads();
                        return true;
                    };
                    if (pred12) closed_form = closed([&](int i, int j) { return Pn12[i + 12 * j]; });
                    else closed_form = closed([&](int i, int j) { return gP[i + (size_t)j * N]; });
                }
            }
            if (!closed_form)
            measurement_moments<NTHREADS>(a, L, bidx, tid, mu, Lp, Z, DZ, Pxz, Sm, zbar, innov, &ish[42],
                                           [&](int t) { return Pin(t, t); }, colbuf, cv.pool - cv.colbuf,
                                           WCHOL && NW == 4 && m <= 8 && a.emit == 0,
                                           [&]() __attribute__((always_inline)) { if constexpr (!LATE_SCAN) ldm_scan(); }, &ldm_on);
            SLK_STAMP(6);
            // removeOutliers (:241 -> :723-754) incl. the shifted second erase (:741-744)
            if (tid == 0) {
                int cnt = m;
                unsigned nout = 0;
                for (int r = 0; r < m; ++r) idx[r] = r;
                int i = 0;
                if (a.gate == 2) {
                    // the caller ran the significance test itself (an arbitrary `mt`, Msckf.hpp:220-223) on the
                    // innovation / covariance of an emit-4 launch: rowsel = { surviving rows, outliers, row indices }
                    const int *rs = a.rowsel + (size_t)bidx * (m + 2);
                    cnt = rs[0] < 0 ? 0 : (rs[0] > m ? m : rs[0]);
                    nout = (unsigned)rs[1];
                    for (int r = 0; r < cnt; ++r) { int v = rs[2 + r]; idx[r] = v < 0 ? 0 : (v >= m ? m - 1 : v); }
                    i = cnt;                                       // skip the built-in loop
                }
                while (i < cnt / 2) {
                    int p = idx[2 * i], q = idx[2 * i + 1];
                    double s00 = Sm[p + m * p], s01 = Sm[p + m * q], s10 = Sm[q + m * p], s11 = Sm[q + m * q];
                    double det = s00 * s11 - s01 * s10, r0 = innov[p], r1 = innov[q];
                    double d2 = (r0 * (s11 * r0 - s01 * r1) + r1 * (s00 * r1 - s10 * r0)) / det;
                    bool ok = a.gate ? (d2 < 5.99) : true;            // chi2_0.95(2), Msckf.hpp:861-865
                    if (!ok) {
                        for (int rep = 0; rep < 2; ++rep) {           // removeRow semantics, :688-697
                            int pos = 2 * i + rep, numRows = cnt - 1;
                            if (pos < numRows) for (int w = pos; w < numRows; ++w) idx[w] = idx[w + 1];
                            cnt = numRows;
                        }
                        nout++;
                    } else {
                        i++;
                    }
                }
                ish[40] = cnt;
                ish[41] = (int)nout;
            }
            __syncthreads();
            SLK_STAMP(7);
            const int mmr = ish[40];
            if (tid == 0 && a.emit != 4) a.outliers[bidx] = (unsigned)ish[41];
            if (a.emit == 4) {
                // innovation and its covariance for a caller-side significance test: Xout [B][m*m + m]
                double *o = a.Xout + (size_t)bidx * (m * m + m);
                for (int e = tid; e < m * m; e += NTHREADS) o[e] = Sm[e];
                for (int e = tid; e < m; e += NTHREADS) o[m * m + e] = innov[e];
            } else if (mmr == 0) {
                status |= SLK_ST_ALL_REJECTED;                         // :250, nothing applied
            } else {
                // K = covXZ * S^-1 (:257).  S = 1/2 dZ dZ^T + R is symmetric positive definite for any
                // valid R: factor it (S = Ls Ls^T) and solve row-wise; a non-SPD S falls back to
                // Gauss-Jordan with partial pivoting (the reference inverts with PartialPivLU).
                // Beside it, the wave that holds the prefix sums turns them into the columns of the factor update
                // (first branch: its 88 live registers must not span the S factorisation of wave 0).
                if (WCHOL && wave == NW - 1) {
                    int st47 = 0;
                    if (ldm_on && mmr <= 8) {
                        unsigned kept = 0;
                        for (int r = 0; r < mmr; ++r) kept |= 1u << idx[r];
                        if constexpr (LATE_SCAN) ldm_scan();
                        const bool pd = ldm_columns(ldm_pf, ldm_a, Sm, m, kept, lane, N, Z, md);
                        st47 = pd ? 1 : 2;
                    }
                    if (lane == 0) ish[47] = st47;
                } else if (wave == 0) {
                    auto sel = [&](int i, int j) { return Sm[idx[i] + m * idx[j]]; };
                    int f0;
                    if (mmr <= 8) {
                        // up to four 2-row blocks: every lane factors the (padded) 8x8 matrix in registers -- no LDS
                        // round trips -- and the lanes t < N solve their row of K straight away (N <= 64: one wave)
                        constexpr int MB = (MST > 0 && MST <= 8) ? MST : 8;      // rows known at compile time: no padding to 8 x 8
                        double gg[MB][MB];
#pragma unroll
                        for (int i = 0; i < MB; ++i)
#pragma unroll
                            for (int j = 0; j <= i; ++j) {
                                const bool in = i < mmr && j < mmr;
                                const double v = Sm[in ? idx[i] + m * idx[j] : 0];
                                gg[i][j] = in ? v : (i == j ? 1.0 : 0.0);
                            }
                        double gi[MB];
                        f0 = -1;
#pragma unroll
                        for (int j = 0; j < MB; ++j) {
                            double d = gg[j][j];
#pragma unroll
                            for (int p = 0; p < j; ++p) d = fma(-gg[j][p], gg[j][p], d);
                            if (f0 < 0 && !(d > 0.0)) f0 = j;
                            double sq, rs;
                            rsqrt_pivot(d, sq, rs);
                            gg[j][j] = sq;
                            gi[j] = rs;
#pragma unroll
                            for (int i = j + 1; i < MB; ++i) {
                                double v = gg[i][j];
#pragma unroll
                                for (int p = 0; p < j; ++p) v = fma(-gg[i][p], gg[j][p], v);
                                gg[i][j] = v * rs;
                            }
                        }
                        // applyDelta's factor as a factor UPDATE (L' = L M, see ldm_columns): needs covXZ = L A, i.e. no
                        // wrapped rotation column; otherwise K is stored and the downdated matrix is factored afresh
                        const bool fastw = ldm_on && f0 < 0;
                        if (f0 < 0) {
                            for (int t = lane; t < N; t += 64) {
                                double x[MB];
#pragma unroll
                                for (int c = 0; c < MB; ++c) {             // forward: Ls w = p
                                    double sum = (c < mmr) ? Pxz[t + N * idx[c < mmr ? c : 0]] : 0.0;
#pragma unroll
                                    for (int p = 0; p < c; ++p) sum = fma(-gg[c][p], x[p], sum);
                                    x[c] = sum * gi[c];
                                }
                                double dsum = 0.0;
#pragma unroll
                                for (int c = MB - 1; c >= 0; --c) {            // backward: Ls^T x = w
                                    double sum = x[c];
#pragma unroll
                                    for (int p = c + 1; p < MB; ++p) sum = fma(-gg[p][c], x[p], sum);
                                    x[c] = sum * gi[c];
                                    if (!fastw && c < mmr) K[t + N * c] = x[c];
                                }
                                if (fastw) {                              // delta = K * innovation (:263), same order as below
#pragma unroll
                                    for (int c = 0; c < MB; ++c) dsum += (c < mmr) ? x[c] * innov[idx[c < mmr ? c : 0]] : 0.0;
                                    delta[t] = dsum;
                                }
                            }
                        }
                    } else if (mmr <= 16) {
                        d4 acc[CholM<1>::NTL];
                        cholm_load<1>(acc, mmr, lane, sel);
                        f0 = cholm_factor<1>(acc, G, mmr, colbuf, lane);
                    } else {
                        d4 acc[CholM<2>::NTL];
                        cholm_load<2>(acc, mmr, lane, sel);
                        f0 = cholm_factor<2>(acc, G, mmr, colbuf, lane);
                    }
                    if (lane == 0) ish[46] = f0;
                }
                __syncthreads();
                const int sfail = ish[46];
                bool singular = false;
                if (sfail < 0 && mmr <= 8) {
                    // K was solved by wave 0 above
                } else if (sfail < 0) {
                    for (int t = tid; t < N; t += NTHREADS) {
                        for (int c = 0; c < mmr; ++c) {               // forward: Ls w = p
                            double sum = Pxz[t + N * idx[c]];
                            for (int p = 0; p < c; ++p) sum -= G[pk(mmr, c, p)] * K[t + N * p];
                            K[t + N * c] = sum / G[pk(mmr, c, c)];
                        }
                        for (int c = mmr - 1; c >= 0; --c) {          // backward: Ls^T x = w
                            double sum = K[t + N * c];
                            for (int p = c + 1; p < mmr; ++p) sum -= G[pk(mmr, p, c)] * K[t + N * p];
                            K[t + N * c] = sum / G[pk(mmr, c, c)];
                        }
                    }
                } else {
                    const int ldj = 2 * mmr + 1;
                    for (int e = tid; e < mmr * mmr; e += NTHREADS) {
                        int r = e % mmr, c = e / mmr;
                        G[r * ldj + c] = Sm[idx[r] + m * idx[c]];
                        G[r * ldj + mmr + c] = (r == c) ? 1.0 : 0.0;
                    }
                    __syncthreads();
                    for (int k = 0; k < mmr; ++k) {
                        int piv = k;
                        double best = fabs(G[k * ldj + k]);
                        for (int i = k + 1; i < mmr; ++i) {
                            double v = fabs(G[i * ldj + k]);
                            if (v > best) { best = v; piv = i; }
                        }
                        if (!(best > 0.0)) { singular = true; break; }
                        __syncthreads();
                        if (piv != k)
                            for (int c = tid; c < 2 * mmr; c += NTHREADS) {
                                double t0 = G[k * ldj + c]; G[k * ldj + c] = G[piv * ldj + c]; G[piv * ldj + c] = t0;
                            }
                        __syncthreads();
                        double pv = G[k * ldj + k];
                        __syncthreads();
                        for (int c = tid; c < 2 * mmr; c += NTHREADS) G[k * ldj + c] = G[k * ldj + c] / pv;
                        __syncthreads();
                        for (int r = tid; r < mmr; r += NTHREADS) {
                            if (r == k) continue;
                            double f = G[r * ldj + k];
                            for (int c = 0; c < 2 * mmr; ++c) G[r * ldj + c] -= f * G[k * ldj + c];
                        }
                        __syncthreads();
                    }
                    if (!singular)
                        for (int e = tid; e < N * mmr; e += NTHREADS) {
                            int t = e % N, c = e / N;
                            double sum = 0.0;
                            for (int c2 = 0; c2 < mmr; ++c2) sum += Pxz[t + N * idx[c2]] * G[c2 * ldj + mmr + c];
                            K[e] = sum;
                        }
                }
                __syncthreads();
                SLK_STAMP(8);
                if (singular) {
                    status |= SLK_ST_SINGULAR;
                } else {
                    // delta = K * innovation (:263)
                    int fastw = 0;                                     // 1 / 2: the factor-update path ran (2: not positive definite)
                    if constexpr (WCHOL) fastw = (mmr <= 8 && sfail < 0) ? ish[47] : 0;
                    if (!fastw)
                    for (int t = tid; t < N; t += NTHREADS) {
                        double sum = 0.0;
                        for (int c = 0; c < mmr; ++c) sum += K[t + N * c] * innov[idx[c]];
                        delta[t] = sum;
                    }
                    SLK_STAMP(9);
                    SLK_STAMP(10);
                    // ---- Pk -= K S K^T (:262) fused into the load of applyDelta's Cholesky (:263 -> :659-662):
                    // K S = covXZ, so the downdated lower triangle is P(i,j) - sum_c covXZ(i,c) K(j,c).
                    if constexpr (WCHOL) {
                        if (fastw == 1) {
                            ldm_product<NT>(Lp, N, DZ, m, Z, md, lane, wave);          // L <- L M on the matrix cores
                            fail = -1;
                        } else if (fastw == 2) {
                            fail = 0;                                  // Pk - K S K^T is not positive definite
                        } else {
                            d4 acc[NT];
                            cholw_load<NT>(acc, N, lane, wave, Pin);
                            cholw_downdate<NT>(acc, N, mmr, lane, wave, [&](int r, int c) { return Pxz[r + N * idx[c]]; },
                                               [&](int r, int c) { return K[r + N * c]; });
                            fail = cholw_factor<NT>(acc, Lp, N, colbuf, md, lane, wave, &ish[45]);
                        }
                    } else if constexpr (NT <= 2 && KST >= 0 && MST > 0) {
                        double xv[MST], kv[MST];                       // this lane's rows of covXZ (kept columns) and of K
#pragma unroll
                        for (int c = 0; c < MST; ++c) {
                            const int row = min(lane, N - 1), cc = c < mmr ? c : 0;
                            xv[c] = (c < mmr) ? Pxz[row + N * idx[cc]] : 0.0;
                            kv[c] = (c < mmr) ? K[row + N * cc] : 0.0;
                        }
                        fail = chol_state([&](int j) {                 // (K(j, c) broadcast from lane j: scalar registers)
                            double sum = 0.0;
#pragma unroll
                            for (int c = 0; c < MST; ++c) sum = fma(xv[c], readlane_f64(kv[c], j), sum);
                            return sum;
                        });
                    } else if constexpr (NT <= 4) {
                        if (wave == 0) {
                            d4 acc[CholM<NT>::NTL];
                            cholm_load<NT>(acc, N, lane, Pin);
                            cholm_downdate<NT>(acc, N, mmr, lane, [&](int r, int c) { return Pxz[r + N * idx[c]]; },
                                               [&](int r, int c) { return K[r + N * c]; });
                            int f0 = cholm_factor<NT>(acc, Lp, N, colbuf, lane);
                            if (lane == 0) ish[45] = f0;
                        }
                        __syncthreads();
                        fail = ish[45];
                    } else {
                        auto down = [&](int i, int j) {
                            double p = Pin(i, j);
                            double sum = 0.0;
                            for (int c = 0; c < mmr; ++c) sum += Pxz[i + N * idx[c]] * K[j + N * c];
                            return p - sum;
                        };
                        if constexpr (BIG) {
                            double *panel = innov + 2 * round_up(m, 2);      // behind the measurement arrays
                            // applyDelta's factor as a factor UPDATE by tile rows (factor_update_blocks): O(N^2 m) instead
                            // of the O(N^3) factorisation of the downdated matrix; needs covXZ = L A (no wrapped rotation
                            // column), at most eight rows and an SPD innovation covariance.  The gain K (in Z's place) is
                            // dead once delta stands: Z holds W and the wave totals, the Cholesky panel the M_JJ tiles.
                            if (m <= 8 && mmr <= 8 && sfail < 0 && ish[42] == 0) {
                                unsigned kept = 0;
                                for (int r = 0; r < mmr; ++r) kept |= 1u << idx[r];
                                __syncthreads();
                                const bool pd = factor_update_blocks<NTHREADS>(Lp, N, DZ, m, Sm, kept, Z, md, Z + round_up(8 * N, 2),
                                                                               panel, &ish[44], tid);
                                fail = pd ? -1 : 0;                          // not positive definite: as a failed LLT
                            } else {
                                fail = chol_blocked_mem<NTHREADS>(Lp, N, panel, colbuf, &ish[45], tid, down);
                            }
                        } else {
                            fail = chol_packed<NTHREADS, SDN>(Lp, N, colbuf, tid, down);
                        }
                    }
                    SLK_STAMP(11);
                    if (fail >= 0) {
                        status |= SLK_ST_LLT_FAIL;
                    } else {
                        redraw = true;
                    }
                }
            }
        }
        if (redraw) {
            // ---- re-drawn sigma points, manifold mean (:664 -> :499-525), covariance (:665)
            const int W = cv.W;
            double *DR = BIG ? a.wsDR + (size_t)bidx * 3 * W : pool;   // rotation deviations, 3 per stored item
            double *Dp = BIG ? pool : pool + round_up(3 * W, 2);       // [2][KP][LDD] panels
            // reference = X[0] = mu + delta (:501)
            for (int t = tid; t < N; t += NTHREADS) {
                int blk = 0, comp = 0, s = t2s(L, t, blk, comp);
                if (s >= 0) ref[s] = mu[s] + delta[t];
            }
            for (int b = tid; b < nso3; b += NTHREADS) {
                const Quat qr = sigma_quat(L, mu, Lp, delta, b, sig_of(0));
                stq(ref + so3_soff(L, b), qr);
                stq(cq + 4 * b, qmul(qconj(qr), ldq(mu + so3_soff(L, b))));
            }
            __syncthreads();
            int it = 0;
            double norm = 0.0;
            bool final_pass = false;
            const bool oe_rebuild = NW == 4 && NT >= 3 && NT <= 4 && a.rebuild_prec == 0;
            for (;;) {                                        // :507-516, then one pass against the final mean (:584)
                // rotation blocks of X_i [-] ref, only the sigma points whose block differs from X_0's
                if constexpr (KST >= 0 && !BIG) {
                    // exact shape: the item count is a compile-time constant -- all rounds of a thread as straight-line
                    // code (index clamped, store predicated) so that their dependency chains interleave
                    constexpr int WST = 6 * KST * KST + 31 * KST + 13;                 // msckf_roff(KST + 1)
                    constexpr int RND = (WST + NTHREADS - 1) / NTHREADS;
                    double dv[RND][3];
#pragma unroll
                    for (int r = 0; r < RND; ++r) {
                        const int w = tid + r * NTHREADS;
                        rot_deviation_desc(a.rtab[w < WST ? w : WST - 1], cq, Lp, delta, dv[r][0], dv[r][1], dv[r][2]);
                    }
#pragma unroll
                    for (int r = 0; r < RND; ++r) {
                        const int w = tid + r * NTHREADS;
                        if (w < WST) { DR[3 * w] = dv[r][0]; DR[3 * w + 1] = dv[r][1]; DR[3 * w + 2] = dv[r][2]; }
                    }
                } else {
                    for (int w = tid; w < W; w += NTHREADS) {
                        double dx, dy, dz;
                        rot_deviation_desc(a.rtab[w], cq, Lp, delta, dx, dy, dz);
                        DR[3 * w] = dx; DR[3 * w + 1] = dy; DR[3 * w + 2] = dz;
                    }
                }
                __syncthreads();
                if (final_pass) break;
                // mean_delta = sum_i (X_i [-] ref) / S.  Vector rows: the +-L_j terms of the pairs
                // cancel, every sigma point contributes (mu + delta) - ref.  Rotation rows: 8 lanes
                // per row over the stored deviations, the other S - cnt points equal X_0's.
                for (int t = tid; t < N; t += NTHREADS) {
                    int blk = 0, comp = 0, s = t2s(L, t, blk, comp);
                    if (s >= 0) md[t] = (mu[s] + delta[t]) - ref[s];
                }
                for (int e = tid / 8; e < 3 * nso3; e += NTHREADS / 8) {
                    const int blk = e / 3, comp = e - 3 * blk, sub = tid & 7;
                    const int r0 = msckf_roff(blk), cnt = msckf_roff(blk + 1) - r0;
                    const double *row = DR + 3 * r0 + comp;
                    double sum;
                    if constexpr (BIG) {
                        // the deviation store is in global memory: sixteen loads in flight per lane, four accumulators
                        double s4[4] = {0.0, 0.0, 0.0, 0.0};
                        for (int i0 = sub; i0 < cnt; i0 += 8 * 16) {
                            double v[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) { const int i = i0 + 8 * u; v[u] = (i < cnt) ? row[3 * i] : 0.0; }
#pragma unroll
                            for (int u = 0; u < 16; ++u) s4[u & 3] += v[u];
                        }
                        sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
                        sum += __shfl_xor(sum, 4, 64);
                        sum += __shfl_xor(sum, 2, 64);
                        sum += __shfl_xor(sum, 1, 64);
                    } else {
                        sum = group_sum<8>(sub, cnt, [&](int i) { return row[3 * i]; });
                    }
                    sum += (double)(S - cnt) * row[0];
                    if (sub == 0) md[msckf_toff(blk) + comp] = sum / (double)S;
                }
                __syncthreads();
                double n2 = group_sum<64>(lane, N, [&](int t) { return md[t] * md[t]; });
                norm = sqrt(n2);
                for (int t = tid; t < N; t += NTHREADS) {       // reference += mean_delta
                    int blk = 0, comp = 0, s = t2s(L, t, blk, comp);
                    if (s >= 0) ref[s] = ref[s] + md[t];
                }
                for (int b = tid; b < nso3; b += NTHREADS) {
                    int to = msckf_toff(b), so = msckf_soff(b);
                    const Quat qr = qmul(ldq(ref + so), so3_exp(md[to], md[to + 1], md[to + 2]));
                    stq(ref + so, qr);
                    stq(cq + 4 * b, qmul(qconj(qr), ldq(mu + so)));
                }
                __syncthreads();
                if (!(norm > 1e-6 && ++it < 10000)) {
                    // The loop leaves with |mean_delta| <= 1e-6 (:511): the deviations against the final mean follow from the
                    // ones just taken by a first-order correction (below, error O(|mean_delta|^2) <= 1e-12) -- the odd / even
                    // rebuild applies it while it pairs the items, the other shapes in a pass of their own.
                    if (it < 10000) break;
                    final_pass = true;
                }
            }
            if (!oe_rebuild && it < 10000) {
                // the same correction for the shapes without the N <= 64 odd / even rebuild, in place (formula: see below);
                // large states: their odd / even rebuild wants each pair as (d+ - d-) / 2, (d+ + d-) / 2 -- same pass
                const bool pairs = BIG && TilePlan<NT, NW>::PASSES == 1;      // (every rebuild precision takes the odd / even form there)
                auto fix = [&](double &x, double &y, double &z, double m0, double m1, double m2) {
                    const double cx = y * m2 - z * m1, cy = z * m0 - x * m2, cz = x * m1 - y * m0;      // d x m
                    const double ax = y * cz - z * cy, ay = z * cx - x * cz, az = x * cy - y * cx;      // d x (d x m)
                    const double a12 = 1.0 / 12.0 + (x * x + y * y + z * z) * (1.0 / 720.0);
                    x = x - m0 + 0.5 * cx - a12 * ax;
                    y = y - m1 + 0.5 * cy - a12 * ay;
                    z = z - m2 + 0.5 * cz - a12 * az;
                };
                for (int w = tid; w < W; w += NTHREADS) {
                    const unsigned lo = (unsigned)a.rtab[w];
                    const unsigned sc = (lo >> 18) & 3u;
                    if (pairs && sc == 2u) continue;                             // a '-' item: done by its '+' partner
                    const int to = (lo >> 20) & 0xff;
                    const double m0 = md[to], m1 = md[to + 1], m2 = md[to + 2];
                    double px = DR[3 * w], py = DR[3 * w + 1], pz = DR[3 * w + 2];
                    fix(px, py, pz, m0, m1, m2);
                    if (pairs && sc == 1u) {
                        double qx = DR[3 * w + 3], qy = DR[3 * w + 4], qz = DR[3 * w + 5];
                        fix(qx, qy, qz, m0, m1, m2);
                        DR[3 * w] = 0.5 * (px - qx); DR[3 * w + 1] = 0.5 * (py - qy); DR[3 * w + 2] = 0.5 * (pz - qz);
                        DR[3 * w + 3] = 0.5 * (px + qx); DR[3 * w + 4] = 0.5 * (py + qy); DR[3 * w + 5] = 0.5 * (pz + qz);
                    } else {
                        DR[3 * w] = px; DR[3 * w + 1] = py; DR[3 * w + 2] = pz;
                    }
                }
                __syncthreads();
            }
            const bool big_oe = BIG && it < 10000;
            if (it >= 10000) status |= SLK_ST_MEAN_NOT_CONVERGED;
            SLK_STAMP(12);
            SLK_NOTE(20, it + 1);
            // mean written out now: `ref` is final
            for (int e = tid; e < Nq; e += NTHREADS) omean[e] = ref[e];
            SLK_STAMP(13);
            // ---- P+ = 1/2 D D^T on the fp64 matrix cores (:665 -> :574-589), D generated panel by
            // panel: vector rows straight from the factor, rotation rows from DR
            constexpr int TN = 16 * NT;
            constexpr int RPT = (TN + 63) / 64;                 // rows of D handled per lane
            constexpr int TPW = TilePlan<NT, NW>::TPW;
            int rkind[RPT], roffs[RPT], rcnt[RPT];
            double rm[RPT], rd[RPT], rr[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                int t = lane + 64 * q, blk = 0, comp = 0;
                rkind[q] = 0; roffs[q] = 0; rcnt[q] = 0; rm[q] = 0.0; rd[q] = 0.0; rr[q] = 0.0;
                if (t < N) {
                    int s = t2s(L, t, blk, comp);
                    if (s >= 0) { rkind[q] = 1; rm[q] = mu[s]; rd[q] = delta[t]; rr[q] = ref[s]; }
                    else { rkind[q] = 2; roffs[q] = 3 * roff[blk] + comp; rcnt[q] = roff[blk + 1] - roff[blk]; }
                } else if (t >= TN) rkind[q] = 3;
            }
            auto gen_panel = [&](int p0, double *Dq) __attribute__((always_inline)) {
                for (int kk = wave; kk < KP; kk += NW) {
                    const int i = p0 + kk;
                    const int j = (i - 1) >> 1;
                    const double sgn = (i & 1) ? 1.0 : -1.0;
                    const int jb = pkcol(N, j);                               // pk(N, t, j) = jb + t
                    // all loads of this column first (branch-free: clamped address, select later)
                    double lv[RPT];
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const int t = lane + 64 * q;
                        const bool vec = rkind[q] == 1 && i > 0 && i < S && j <= t;
                        const bool rot = rkind[q] == 2 && i < S;
                        const double *src = rot ? DR + (roffs[q] + 3 * (i < rcnt[q] ? i : 0)) : Lp + (vec ? jb + t : 0);
                        lv[q] = *src;
                    }
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const int t = lane + 64 * q;
                        double v = 0.0;
                        if (i < S) {
                            if (rkind[q] == 1) {
                                const double l = (i > 0 && j <= t) ? sgn * lv[q] : 0.0;
                                v = (rm[q] + (rd[q] + l)) - rr[q];
                            } else if (rkind[q] == 2) {
                                v = lv[q];
                            }
                        }
                        if (rkind[q] != 3) Dq[kk * LDD + t] = v;
                    }
                }
            };
            // the same in two halves for the fp64 panel loop: the loads of the NEXT panel are issued before the MFMA block of
            // the current one (the factor / the deviations may live in the global workspace) and stored to LDS after it
            constexpr int KPW = (KP + NW - 1) / NW;                               // panel columns per wave
            auto gen_load = [&](int p0, double (&lv)[KPW][RPT]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < KPW; ++u) {
                    const int kk = wave + u * NW, i = p0 + kk, j = (i - 1) >> 1;
                    const int jb = pkcol(N, j);
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const int t = lane + 64 * q;
                        const bool vec = rkind[q] == 1 && i > 0 && i < S && j <= t && kk < KP;
                        const bool rot = rkind[q] == 2 && i < S && kk < KP;
                        const double *src = rot ? DR + (roffs[q] + 3 * (i < rcnt[q] ? i : 0)) : Lp + (vec ? jb + t : 0);
                        lv[u][q] = *src;
                    }
                }
            };
            auto gen_store = [&](int p0, const double (&lv)[KPW][RPT], double *Dq) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < KPW; ++u) {
                    const int kk = wave + u * NW, i = p0 + kk, j = (i - 1) >> 1;
                    const double sgn = (i & 1) ? 1.0 : -1.0;
                    if (kk < KP) {
#pragma unroll
                        for (int q = 0; q < RPT; ++q) {
                            const int t = lane + 64 * q;
                            double v = 0.0;
                            if (i < S) {
                                if (rkind[q] == 1) {
                                    const double l = (i > 0 && j <= t) ? sgn * lv[u][q] : 0.0;
                                    v = (rm[q] + (rd[q] + l)) - rr[q];
                                } else if (rkind[q] == 2) {
                                    v = lv[u][q];
                                }
                            }
                            if (rkind[q] != 3) Dq[kk * LDD + t] = v;
                        }
                    }
                }
            };
            bool rebuilt = false;
            if constexpr (NW == 4 && NT >= 3 && NT <= 4) {
                if (a.rebuild_prec == 0) {
                    // ---- Odd / even rebuild.  The sigma points come in pairs X(2j+1), X(2j+2) = mu [+] (delta +- L_j):
                    //     1/2 sum_i d_i d_i^T = 1/2 d_0 d_0^T + sum_j (o_j o_j^T + e_j e_j^T),   o = (d+ - d-)/2, e = (d+ + d-)/2.
                    // Vector rows: o = L'(:, j), e = 0 (exactly, up to the rounding of mu + delta).  Rotation rows: the odd /
                    // even parts of the stored deviations; beyond the block's columns (j > toff + 2) o = 0 and e = d_0.
                    // So P+ = O O^T + E E^T + 1/2 d_0 d_0^T with HALF the k-steps of 1/2 D D^T for the N x N part, fragments
                    // read as they lie in the packed factor / the deviation store, and E E^T a 27 x 27 problem in the index
                    // space of the rotation rows (3 tiles), added where the tiles leave for memory.
                    // The stored deviations are against the reference BEFORE its last move m = mean_delta (|m| <= 1e-6): against
                    // the final mean  d' = log(exp(-m) exp(d)) = d - Jl^-1(d) m + O(|m|^2),
                    //     Jl^-1(d) m = m - 1/2 d x m + a d x (d x m),  a = 1/|d|^2 - (1 + cos|d|) / (2 |d| sin|d|) = 1/12 + |d|^2/720 + ...
                    // (left Jacobian of SO(3); the next terms are below 1e-12 for the rotations a sigma point spreads).  After a
                    // non-converged loop (max_it, status flagged) the stored deviations are already the final pass.
                    const bool corr = it < 10000;
                    for (int w = tid; w < W; w += NTHREADS) {
                        const unsigned lo = (unsigned)a.rtab[w];
                        const unsigned sc = (lo >> 18) & 3u;
                        if (sc == 2u) continue;                                  // a '-' item: done by its '+' partner
                        const int to = (lo >> 20) & 0xff;
                        const double m0 = corr ? md[to] : 0.0, m1 = corr ? md[to + 1] : 0.0, m2 = corr ? md[to + 2] : 0.0;
                        auto fix = [&](double &x, double &y, double &z) {
                            const double cx = y * m2 - z * m1, cy = z * m0 - x * m2, cz = x * m1 - y * m0;      // d x m
                            const double ax = y * cz - z * cy, ay = z * cx - x * cz, az = x * cy - y * cx;      // d x (d x m)
                            const double a12 = 1.0 / 12.0 + (x * x + y * y + z * z) * (1.0 / 720.0);
                            x = x - m0 + 0.5 * cx - a12 * ax;
                            y = y - m1 + 0.5 * cy - a12 * ay;
                            z = z - m2 + 0.5 * cz - a12 * az;
                        };
                        double px = DR[3 * w], py = DR[3 * w + 1], pz = DR[3 * w + 2];
                        fix(px, py, pz);
                        if (sc == 0u) {                                          // the centre point
                            DR[3 * w] = px; DR[3 * w + 1] = py; DR[3 * w + 2] = pz;
                        } else {                                                 // a '+' item: its '-' partner follows
                            double qx = DR[3 * w + 3], qy = DR[3 * w + 4], qz = DR[3 * w + 5];
                            fix(qx, qy, qz);
                            DR[3 * w] = 0.5 * (px - qx); DR[3 * w + 1] = 0.5 * (py - qy); DR[3 * w + 2] = 0.5 * (pz - qz);
                            DR[3 * w + 3] = 0.5 * (px + qx); DR[3 * w + 4] = 0.5 * (py + qy); DR[3 * w + 5] = 0.5 * (pz + qz);
                        }
                    }
                    if (tid == 0) md[0] = 0.0;                                  // the zero every masked fragment reads
                    __syncthreads();
                    constexpr int NTL = CholM<NT>::NTL;
                    constexpr int HALF = (NTL + 1) / 2;                         // tiles per tile group
                    const int th = wave >> 1, kh = wave & 1;                    // tile group, k-step group
                    const int c16 = lane & 15, g4 = lane >> 4;
                    const char *lds = reinterpret_cast<const char *>(smem);
                    const int LpB = 8 * (int)(Lp - smem), DRB = 8 * (int)(DR - smem), zeroB = 8 * (int)(md - smem);
                    int baseB[NT], thr[NT];
                    bool isv[NT];
#pragma unroll
                    for (int I = 0; I < NT; ++I) {
                        const int tr = 16 * I + c16;
                        int blk = 0, comp = 0;
                        isv[I] = true; thr[I] = -1; baseB[I] = zeroB;
                        if (tr < N) {
                            const int s = t2s(L, tr, blk, comp);
                            if (s >= 0) { thr[I] = tr; baseB[I] = LpB + 8 * tr; }
                            else { isv[I] = false; thr[I] = msckf_toff(blk) + 2; baseB[I] = DRB + 8 * (3 * msckf_roff(blk) + comp + 3); }
                        }
                    }
                    d4 acc[HALF];
#pragma unroll
                    for (int q = 0; q < HALF; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
                    const int nks = (N + 3) >> 2;
                    constexpr int ROWS_A = TileMap<NT>::row(HALF - 1) + 1;       // tile rows the first tile group touches
                    for (int ks = kh; ks < nks; ks += 2) {
                        const int j = 4 * ks + g4;
                        const int varV = 8 * pkcol(N, j), varR = 48 * j;
                        double frag[NT];
#pragma unroll
                        for (int I = 0; I < NT; ++I) {
                            if (I >= ROWS_A && th == 0) { frag[I] = 0.0; continue; }
                            const bool c = j <= thr[I];
                            const int ad = c ? baseB[I] + (isv[I] ? varV : varR) : zeroB;
                            frag[I] = *reinterpret_cast<const double *>(lds + ad);
                        }
#pragma unroll
                        for (int I = 0; I < NT; ++I)
#pragma unroll
                            for (int J = 0; J <= I; ++J)
                                if (tile_idx(I, J) / HALF == th)
                                    acc[tile_idx(I, J) % HALF] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                        frag[I], frag[J], acc[tile_idx(I, J) % HALF], 0, 0, 0);
                    }
                    // E E^T + 1/2 d_0 d_0^T in the index space of the rotation rows: rho = 3 b + comp, tiles (0,0) (1,0) (1,1)
                    // by waves 0, 1, 2
                    d4 accE = {0.0, 0.0, 0.0, 0.0};
                    if (wave < 3) {
                        const int Ir = wave >= 1 ? 1 : 0, Ic = wave == 2 ? 1 : 0;
                        const int nrot = 3 * nso3;
                        int bE[2], tE[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int rho = 16 * (u ? Ic : Ir) + c16;
                            const int b = rho / 3, comp = rho - 3 * b;
                            const bool okr = rho < nrot;
                            bE[u] = okr ? DRB + 8 * (3 * msckf_roff(okr ? b : 0) + comp) : -1;
                            tE[u] = msckf_toff(okr ? b : 0) + 2;
                        }
                        for (int ks = 0; ks < nks; ++ks) {
                            const int j = 4 * ks + g4;
                            double fe[2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int ad = (bE[u] >= 0 && j < N) ? bE[u] + (j <= tE[u] ? 48 * j + 48 : 0) : zeroB;
                                fe[u] = *reinterpret_cast<const double *>(lds + ad);
                            }
                            accE = __builtin_amdgcn_mfma_f64_16x16x4f64(fe[0], fe[1], accE, 0, 0, 0);
                        }
                        double f0[2];                                            // the centre point: 1/2 d_0 d_0^T
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const double d0 = *reinterpret_cast<const double *>(lds + ((bE[u] >= 0 && g4 == 0) ? bE[u] : zeroB));
                            f0[u] = d0 * 0.70710678118654752440;
                        }
                        accE = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[0], f0[1], accE, 0, 0, 0);
                    }
                    __syncthreads();                     // factor, deviations and vectors are dead from here
                    SLK_STAMP(14);
                    constexpr int TS = 16 * 17;          // padded 16x16 tile, [col][row]
                    double *EEt = smem + (cv.total - 3 * TS);
                    int TR = (cv.total - 3 * TS) / (2 * TS);
                    if (TR > NTL) TR = NTL;
                    double *red = smem;
                    if (wave < 3) {
                        double *dst = EEt + wave * TS + c16 * 17 + g4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dst[4 * q] = accE[q];
                    }
                    const int ea = tid & 15, eb = (tid >> 4) & 15;
                    // rotation-row index of a tangent row (-1: a vector row or padding): State.hpp:141-149, :246-252, :384-396
                    auto rho_of = [&](int tr) -> int {
                        if (tr >= N) return -1;
                        if (tr < 12) return (tr >= 3 && tr < 6) ? tr - 3 : -1;
                        const int cc = (tr - 12) / 6, r = (tr - 12) - 6 * cc;
                        return r >= 3 ? 3 + 3 * cc + (r - 3) : -1;
                    };
                    int rra[NT], rrb[NT];
#pragma unroll
                    for (int I = 0; I < NT; ++I) { rra[I] = rho_of(16 * I + ea); rrb[I] = rho_of(16 * I + eb); }
                    auto ee_val = [&](int r1, int r2) -> double {
                        const int hi = r1 > r2 ? r1 : r2, lo = r1 > r2 ? r2 : r1;
                        const bool in = lo >= 0;
                        const double v = EEt[in ? ((hi >> 4) + (lo >> 4)) * TS + (lo & 15) * 17 + (hi & 15) : 0];
                        return in ? v : 0.0;
                    };
                    for (int T0 = 0; T0 < NTL; T0 += TR) {
#pragma unroll
                        for (int T = 0; T < NTL; ++T)
                            if (T / HALF == th && T >= T0 && T < T0 + TR) {
                                double *dst = red + (kh * TR + (T - T0)) * TS + c16 * 17 + g4;
#pragma unroll
                                for (int q = 0; q < 4; ++q) dst[4 * q] = acc[T % HALF][q];
                            }
                        __syncthreads();
                        const int ntl = (NTL - T0 < TR) ? NTL - T0 : TR;
                        for (int Tl = 0; Tl < ntl; ++Tl) {                       // 256 threads per tile; tile index uniform
                            const int T = T0 + Tl;
                            const int I = (T >= 1) + (T >= 3) + (T >= 6), J = T - I * (I + 1) / 2;
                            const double *src = red + Tl * TS;
                            int ra = rra[0], rbI = rrb[0], raJ = rra[0], rb = rrb[0];
#pragma unroll
                            for (int q = 1; q < NT; ++q) {
                                if (I == q) { ra = rra[q]; rbI = rrb[q]; }
                                if (J == q) { raJ = rra[q]; rb = rrb[q]; }
                            }
                            {   // lower triangle: consecutive lanes = consecutive rows of one column
                                const double sum = src[eb * 17 + ea] + src[TR * TS + eb * 17 + ea] + ee_val(ra, rb);
                                const int row = 16 * I + ea, col = 16 * J + eb;
                                if (row < N && col < N) oP[row + (size_t)col * N] = sum;
                            }
                            if (I != J) {   // mirrored copy, again contiguous in the fast index
                                const double sum = src[ea * 17 + eb] + src[TR * TS + ea * 17 + eb] + ee_val(rbI, raJ);
                                const int row = 16 * I + eb, col = 16 * J + ea;
                                if (row < N && col < N) oP[col + (size_t)row * N] = sum;
                            }
                        }
                        __syncthreads();
                    }
                    rebuilt = true;
                }
            }
            if constexpr (NT <= 4) {
                if (!rebuilt && a.rebuild_prec == 0) {
                    // ---- K-split rebuild: every wave owns ALL lower tiles for its share of the sigma
                    // points (k-steps dealt round-robin), builds its A/B fragments straight from the
                    // packed factor / the rotation deviations -- no panel staging, no barrier, no
                    // fragment computed twice -- and the partial tiles are summed through LDS, which
                    // also lets both triangles go out as contiguous 128-byte rows.
                    constexpr int NTL = CholM<NT>::NTL;
                    // four waves = 2 halves of the sigma points x 2 halves of the tile list: 5 accumulator
                    // tiles per wave instead of 10 (registers), two partial sums per tile instead of four
                    constexpr int TSPLIT = (NW == 4) ? 2 : 1;          // tile groups
                    constexpr int KSPLIT = NW / TSPLIT;                // sigma-point groups
                    constexpr int HALF = (NTL + TSPLIT - 1) / TSPLIT;  // tiles per group
                    const int th = (TSPLIT == 2) ? (wave >> 1) : 0, kh = (TSPLIT == 2) ? (wave & 1) : wave;
                    d4 acc[HALF];
#pragma unroll
                    for (int q = 0; q < HALF; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
                    const int c16 = lane & 15, g4 = lane >> 4;
                    // Row t = 16 I + c16 of D, per lane and tile row, in ONE branch-free form
                    //   D(t, i) = (qm + (qd + f * smem[bas + (c ? y : 0)])) - qr
                    // vector row : bas = factor, c = j < t + 1, y = pk(N, t, j), f = c ? sgn : 0
                    // rotation   : bas = its deviations, c = i < count, y = 3 i, f = 1, qm = qd = qr = 0
                    // padding    : a vector row with threshold 0 and zero constants
                    bool isv[NT];
                    int bas[NT], thr[NT];
                    double qm[NT], qd[NT], qr[NT];
                    const int LpOff = (int)(Lp - smem), DROff = (int)(DR - smem);
#pragma unroll
                    for (int I = 0; I < NT; ++I) {
                        int t = 16 * I + c16, blk = 0, comp = 0;
                        isv[I] = true; bas[I] = LpOff; thr[I] = 0; qm[I] = 0.0; qd[I] = 0.0; qr[I] = 0.0;
                        if (t < N) {
                            int s = t2s(L, t, blk, comp);
                            if (s >= 0) { thr[I] = t + 1; qm[I] = mu[s]; qd[I] = delta[t]; qr[I] = ref[s]; }
                            else {
                                isv[I] = false;
                                bas[I] = DROff + 3 * msckf_roff(blk) + comp;
                                thr[I] = msckf_roff(blk + 1) - msckf_roff(blk);
                            }
                        }
                    }
                    const int nks = (S + 3) >> 2;
                    for (int ks = kh; ks < nks; ks += KSPLIT) {
                        const int i = 4 * ks + g4;
                        const bool valid = i < S;
                        const int j = (i > 0) ? ((i - 1) >> 1) : 0;
                        const double sgn = (i == 0) ? 0.0 : ((i & 1) ? 1.0 : -1.0);
                        const int jbc = pkcol(N, j) + c16;                        // pk(N, t, j) = jbc + 16 I
                        const int i3 = 3 * i;
                        double frag[NT];
                        constexpr int ROWS_A = TileMap<NT>::row(HALF - 1) + 1;   // tile rows the first tile group touches
#pragma unroll
                        for (int I = 0; I < NT; ++I) {
                            if (TSPLIT == 2 && I >= ROWS_A && th == 0) { frag[I] = 0.0; continue; }
                            const bool c = (isv[I] ? j : i) < thr[I];
                            const int y = isv[I] ? jbc + 16 * I : i3;
                            const double l = smem[bas[I] + (c ? y : 0)];
                            const double f = isv[I] ? (c ? sgn : 0.0) : 1.0;
                            const double v = (qm[I] + (qd[I] + f * l)) - qr[I];
                            frag[I] = valid ? v : 0.0;
                        }
#pragma unroll
                        for (int I = 0; I < NT; ++I)
#pragma unroll
                            for (int J = 0; J <= I; ++J)
                                if (tile_idx(I, J) / HALF == th)
                                    acc[tile_idx(I, J) % HALF] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                        frag[I], frag[J], acc[tile_idx(I, J) % HALF], 0, 0, 0);
                    }
                    __syncthreads();                     // factor, deviations and vectors are dead from here
                    SLK_STAMP(14);
                    constexpr int TS = 16 * 17;          // padded 16x16 tile, [col][row]
                    int TR = cv.total / (KSPLIT * TS);
                    if (TR > NTL) TR = NTL;
                    double *red = smem;
                    const int ea = tid & 15, eb = (tid >> 4) & 15;
                    for (int T0 = 0; T0 < NTL; T0 += TR) {
#pragma unroll
                        for (int T = 0; T < NTL; ++T)
                            if (T / HALF == th && T >= T0 && T < T0 + TR) {
                                double *dst = red + (kh * TR + (T - T0)) * TS + c16 * 17 + g4;
#pragma unroll
                                for (int q = 0; q < 4; ++q) dst[4 * q] = acc[T % HALF][q];
                            }
                        __syncthreads();
                        const int ntl = (NTL - T0 < TR) ? NTL - T0 : TR;
                        // 256 threads per tile (the 64-thread kernels take four strides); tile index uniform
                        for (int Tl = 0; Tl < ntl; ++Tl) {
                            const int T = T0 + Tl;
                            const int I = (T >= 1) + (T >= 3) + (T >= 6), J = T - I * (I + 1) / 2;
                            const double *src = red + Tl * TS;
                            for (int e2 = eb; e2 < 16; e2 += NTHREADS / 16) {
                                {   // lower triangle: consecutive lanes = consecutive rows of one column
                                    double sum = 0.0;
#pragma unroll
                                    for (int w = 0; w < KSPLIT; ++w) sum += src[w * TR * TS + e2 * 17 + ea];
                                    const int row = 16 * I + ea, col = 16 * J + e2;
                                    if (row < N && col < N) oP[row + (size_t)col * N] = 0.5 * sum;
                                }
                                if (I != J) {   // mirrored copy, again contiguous in the fast index
                                    double sum = 0.0;
#pragma unroll
                                    for (int w = 0; w < KSPLIT; ++w) sum += src[w * TR * TS + ea * 17 + e2];
                                    const int row = 16 * I + e2, col = 16 * J + ea;
                                    if (row < N && col < N) oP[col + (size_t)row * N] = 0.5 * sum;
                                }
                            }
                        }
                        __syncthreads();
                    }
                    rebuilt = true;
                }
            }
            if constexpr (BIG && TilePlan<NT, NW>::PASSES == 1) {
                if (big_oe) {
                    // ---- Odd / even rebuild for the large states, through the LDS panels: P+ = O O^T + E E^T + 1/2 d_0 d_0^T
                    // (see the N <= 64 version above).  A panel column is now a COLUMN j of the pair structure: O(:, j) =
                    // L'(:, j) for the vector rows (read as it lies in the factor) and the odd parts of the stored pairs
                    // for the rotation rows -- N + 1 columns instead of the 2 N + 1 sigma points; E lives in the index
                    // space of the rotation rows (rho = 3 b + comp), its column j the even parts (d_0 beyond the block's
                    // own columns), column N = d_0 / sqrt(2); its tiles are added to the covariance where they belong.
                    constexpr int NTE = (3 * (1 + (16 * NT - 12) / 6) + 15) / 16;         // tile rows of E
                    constexpr int LDE = 16 * NTE;
                    constexpr int TPWE = TilePlan<NTE, NW>::TPW;
                    static_assert(TilePlan<NTE, NW>::PASSES == 1, "E tiles in one pass");
                    double *Ep = Dp + 2 * KP * LDD;                                    // [2][KP][LDE]
                    const int nrot = 3 * nso3, NC = N + 1;
                    auto gen_oe_load = [&](int j0, double (&lv)[KPW][RPT], double (&ev)[KPW][2]) __attribute__((always_inline)) {
#pragma unroll
                        for (int u = 0; u < KPW; ++u) {
                            const int kk = wave + u * NW, j = j0 + kk;
                            const int jb = pkcol(N, j < N ? j : 0);
#pragma unroll
                            for (int q = 0; q < RPT; ++q) {
                                const int t = lane + 64 * q;
                                const bool vec = rkind[q] == 1 && j < N && j <= t && kk < KP;
                                const bool rot = rkind[q] == 2 && j < N && 1 + 2 * j < rcnt[q] && kk < KP;
                                const double *src = rot ? DR + (roffs[q] + 3 * (1 + 2 * j)) : Lp + (vec ? jb + t : 0);
                                const double v = *src;
                                lv[u][q] = (vec || rot) ? v : 0.0;
                            }
#pragma unroll
                            for (int qe = 0; qe < 2; ++qe) {
                                const int rho = lane + 64 * qe;
                                const bool in = rho < nrot && j < NC && kk < KP;
                                const int b = in ? rho / 3 : 0, comp = rho - 3 * b;
                                const int cnt = msckf_roff(b + 1) - msckf_roff(b), base = 3 * msckf_roff(b) + comp;
                                const bool pair = j < N && 2 + 2 * j < cnt;
                                const double v = DR[in ? base + (pair ? 3 * (2 + 2 * j) : 0) : 0];
                                ev[u][qe] = in ? (j == N ? v * 0.70710678118654752440 : v) : 0.0;
                            }
                        }
                    };
                    auto gen_oe_store = [&](const double (&lv)[KPW][RPT], const double (&ev)[KPW][2], double *Dq, double *Eq)
                        __attribute__((always_inline)) {
#pragma unroll
                        for (int u = 0; u < KPW; ++u) {
                            const int kk = wave + u * NW;
                            if (kk < KP) {
#pragma unroll
                                for (int q = 0; q < RPT; ++q) {
                                    const int t = lane + 64 * q;
                                    if (rkind[q] != 3) Dq[kk * LDD + t] = lv[u][q];
                                }
#pragma unroll
                                for (int qe = 0; qe < 2; ++qe) {
                                    const int rho = lane + 64 * qe;
                                    if (rho < LDE) Eq[kk * LDE + rho] = ev[u][qe];
                                }
                            }
                        }
                    };
                    double *EEt = pool;             // E E^T (+ 1/2 d_0 d_0^T) meets the O O^T tiles in LDS once the panels are dead
                    if (a.rebuild_prec == 0) {
                    d4 acc[TPW], accE[TPWE];
#pragma unroll
                    for (int q = 0; q < TPW; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < TPWE; ++q) accE[q] = d4{0.0, 0.0, 0.0, 0.0};
                    double lvn[KPW][RPT], evn[KPW][2];
                    gen_oe_load(0, lvn, evn);
                    gen_oe_store(lvn, evn, Dp, Ep);
                    __syncthreads();
                    int pb = 0;
                    for (int j0 = 0; j0 < NC; j0 += KP, pb ^= 1) {
                        const double *Dc = Dp + pb * KP * LDD, *Ec = Ep + pb * KP * LDE;
                        const bool more = j0 + KP < NC;
                        if (more) gen_oe_load(j0 + KP, lvn, evn);            // the next panel's operands: in flight under the MFMAs
#pragma unroll
                        for (int ks = 0; ks < KP / 4; ++ks) {
                            double frag[NT], fragE[NTE];
#pragma unroll
                            for (int I = 0; I < NT; ++I) frag[I] = Dc[(4 * ks + (lane >> 4)) * LDD + 16 * I + (lane & 15)];
#pragma unroll
                            for (int I = 0; I < NTE; ++I) fragE[I] = Ec[(4 * ks + (lane >> 4)) * LDE + 16 * I + (lane & 15)];
                            MfmaTiles<NT, NW, 0>::run(frag, acc, wave, 0);
                            MfmaTiles<NTE, NW, 0>::run(fragE, accE, wave, 0);
                        }
                        if (more) gen_oe_store(lvn, evn, Dp + (pb ^ 1) * KP * LDD, Ep + (pb ^ 1) * KP * LDE);
                        __syncthreads();
                    }
                    SLK_STAMP(14);
                    // (lower triangle of tiles, 16 x 17 each; every stored element picks up its own)
                    MfmaTiles<NTE, NW, 0>::to_lds(EEt, accE, wave, lane);
                    __syncthreads();
                    MfmaTiles<NT, NW, 0>::store_plus_rot(oP, N, acc, wave, lane, EEt);
                    } else if (a.rebuild_prec == 1) {
                    // ---- BASELINE config 5, fp32: the same panels, operands rounded to fp32 where they leave LDS,
                    // v_mfma_f32_16x16x4_f32 with fp32 accumulation (half the matrix time of fp64, half the accumulator registers)
                    f4 acc[TPW], accE[TPWE];
#pragma unroll
                    for (int q = 0; q < TPW; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < TPWE; ++q) accE[q] = f4{0.f, 0.f, 0.f, 0.f};
                    double lvn[KPW][RPT], evn[KPW][2];
                    gen_oe_load(0, lvn, evn);
                    gen_oe_store(lvn, evn, Dp, Ep);
                    __syncthreads();
                    int pb = 0;
                    for (int j0 = 0; j0 < NC; j0 += KP, pb ^= 1) {
                        const double *Dc = Dp + pb * KP * LDD, *Ec = Ep + pb * KP * LDE;
                        const bool more = j0 + KP < NC;
                        if (more) gen_oe_load(j0 + KP, lvn, evn);
#pragma unroll
                        for (int ks = 0; ks < KP / 4; ++ks) {
                            float frag[NT], fragE[NTE];
#pragma unroll
                            for (int I = 0; I < NT; ++I) frag[I] = (float)Dc[(4 * ks + (lane >> 4)) * LDD + 16 * I + (lane & 15)];
#pragma unroll
                            for (int I = 0; I < NTE; ++I) fragE[I] = (float)Ec[(4 * ks + (lane >> 4)) * LDE + 16 * I + (lane & 15)];
                            MfmaTiles32<NT, NW, 0>::run(frag, acc, wave, 0);
                            MfmaTiles32<NTE, NW, 0>::run(fragE, accE, wave, 0);
                        }
                        if (more) gen_oe_store(lvn, evn, Dp + (pb ^ 1) * KP * LDD, Ep + (pb ^ 1) * KP * LDE);
                        __syncthreads();
                    }
                    SLK_STAMP(14);
                    MfmaTiles32<NTE, NW, 0>::to_lds(EEt, accE, wave, lane);
                    __syncthreads();
                    MfmaTiles32<NT, NW, 0>::store_plus_rot(oP, N, acc, wave, lane, EEt);
                    } else {
                    // ---- BASELINE config 5, bf16 operands / fp32 accumulation: panels of 32 pair columns written as bf16
                    // [row][32] (a lane's eight k-values are contiguous), v_mfma_f32_16x16x32_bf16: one matrix instruction per
                    // tile and 32 columns instead of eight
                    f4 acc[TPW], accE[TPWE];
#pragma unroll
                    for (int q = 0; q < TPW; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < TPWE; ++q) accE[q] = f4{0.f, 0.f, 0.f, 0.f};
                    __bf16 *Db = reinterpret_cast<__bf16 *>(Dp), *Eb = Db + TN * 32;      // [TN][32], [LDE][32]
                    for (int j0 = 0; j0 < NC; j0 += 32) {
                        for (int kk = wave; kk < 32; kk += NW) {
                            const int j = j0 + kk;
                            const int jb = pkcol(N, j < N ? j : 0);
#pragma unroll
                            for (int q = 0; q < RPT; ++q) {
                                const int t = lane + 64 * q;
                                const bool vec = rkind[q] == 1 && j < N && j <= t;
                                const bool rot = rkind[q] == 2 && j < N && 1 + 2 * j < rcnt[q];
                                const double *src = rot ? DR + (roffs[q] + 3 * (1 + 2 * j)) : Lp + (vec ? jb + t : 0);
                                const double v = *src;
                                if (t < TN) Db[t * 32 + kk] = (__bf16)(float)((vec || rot) ? v : 0.0);
                            }
#pragma unroll
                            for (int qe = 0; qe < 2; ++qe) {
                                const int rho = lane + 64 * qe;
                                const bool in = rho < nrot && j < NC;
                                const int b = in ? rho / 3 : 0, comp = rho - 3 * b;
                                const int cnt = msckf_roff(b + 1) - msckf_roff(b), base = 3 * msckf_roff(b) + comp;
                                const bool pair = j < N && 2 + 2 * j < cnt;
                                const double v = DR[in ? base + (pair ? 3 * (2 + 2 * j) : 0) : 0];
                                if (rho < LDE) Eb[rho * 32 + kk] = (__bf16)(float)(in ? (j == N ? v * 0.70710678118654752440 : v) : 0.0);
                            }
                        }
                        __syncthreads();
                        b8 frag[NT], fragE[NTE];
#pragma unroll
                        for (int I = 0; I < NT; ++I) frag[I] = *reinterpret_cast<const b8 *>(Db + (16 * I + (lane & 15)) * 32 + 8 * (lane >> 4));
#pragma unroll
                        for (int I = 0; I < NTE; ++I) fragE[I] = *reinterpret_cast<const b8 *>(Eb + (16 * I + (lane & 15)) * 32 + 8 * (lane >> 4));
                        MfmaTiles32<NT, NW, 0>::run_bf16(frag, acc, wave, 0);
                        MfmaTiles32<NTE, NW, 0>::run_bf16(fragE, accE, wave, 0);
                        __syncthreads();
                    }
                    SLK_STAMP(14);
                    MfmaTiles32<NTE, NW, 0>::to_lds(EEt, accE, wave, lane);
                    __syncthreads();
                    MfmaTiles32<NT, NW, 0>::store_plus_rot(oP, N, acc, wave, lane, EEt);
                    }
                    rebuilt = true;
                }
            }
            // (large states take the odd / even form above at every precision; after a mean that did not converge -- status
            // flagged -- they rebuild in fp64 whatever the mode)
            if (rebuilt) {
            } else if (a.rebuild_prec == 0 || BIG) {
            for (int pass = 0; pass < TilePlan<NT, NW>::PASSES; ++pass) {
                d4 acc[TPW];
#pragma unroll
                for (int q = 0; q < TPW; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
                gen_panel(0, Dp);
                __syncthreads();
                int pb = 0;
                for (int p0 = 0; p0 < S; p0 += KP, pb ^= 1) {
                    const double *Dc = Dp + pb * KP * LDD;
                    double lnext[KPW][RPT];
                    const bool more = p0 + KP < S;
                    if (more) gen_load(p0 + KP, lnext);
#pragma unroll
                    for (int ks = 0; ks < KP / 4; ++ks) {
                        double frag[NT];
#pragma unroll
                        for (int I = 0; I < NT; ++I) frag[I] = Dc[(4 * ks + (lane >> 4)) * LDD + 16 * I + (lane & 15)];
                        MfmaTiles<NT, NW, 0>::run(frag, acc, wave, pass);
                    }
                    if (more) gen_store(p0 + KP, lnext, Dp + (pb ^ 1) * KP * LDD);
                    __syncthreads();
                }
                SLK_STAMP(14);
                MfmaTiles<NT, NW, 0>::store(oP, N, acc, wave, lane, pass);
            }
            } else if (a.rebuild_prec == 1) {
                // ---- precision sweep: fp32 operands and accumulation
                for (int pass = 0; pass < TilePlan<NT, NW>::PASSES; ++pass) {
                    f4 acc[TPW];
#pragma unroll
                    for (int q = 0; q < TPW; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
                    gen_panel(0, Dp);
                    __syncthreads();
                    int pb = 0;
                    for (int p0 = 0; p0 < S; p0 += KP, pb ^= 1) {
                        const double *Dc = Dp + pb * KP * LDD;
#pragma unroll
                        for (int ks = 0; ks < KP / 4; ++ks) {
                            float frag[NT];
#pragma unroll
                            for (int I = 0; I < NT; ++I)
                                frag[I] = (float)Dc[(4 * ks + (lane >> 4)) * LDD + 16 * I + (lane & 15)];
                            MfmaTiles32<NT, NW, 0>::run(frag, acc, wave, pass);
                        }
                        if (p0 + KP < S) gen_panel(p0 + KP, Dp + (pb ^ 1) * KP * LDD);
                        __syncthreads();
                    }
                    MfmaTiles32<NT, NW, 0>::store(oP, N, acc, wave, lane, pass);
                }
            } else {
                // ---- precision sweep: bf16 operands, fp32 accumulation, 32 sigma points per MFMA.
                // Panel = bf16 [16*NT rows][32 sigma points] (64-byte rows), single-buffered.
                __bf16 *Db = reinterpret_cast<__bf16 *>(Dp);
                for (int pass = 0; pass < TilePlan<NT, NW>::PASSES; ++pass) {
                    f4 acc[TPW];
#pragma unroll
                    for (int q = 0; q < TPW; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
                    for (int p0 = 0; p0 < S; p0 += 32) {
                        for (int kk = wave; kk < 32; kk += NW) {
                            const int i = p0 + kk;
                            const int j = (i - 1) >> 1;
                            const double sgn = (i & 1) ? 1.0 : -1.0;
#pragma unroll
                            for (int q = 0; q < RPT; ++q) {
                                const int t = lane + 64 * q;
                                double v = 0.0;
                                if (i < S) {
                                    if (rkind[q] == 1) {
                                        double l = (i > 0 && j <= t) ? sgn * Lp[pk(N, t, j)] : 0.0;
                                        v = (rm[q] + (rd[q] + l)) - rr[q];
                                    } else if (rkind[q] == 2) {
                                        v = DR[roffs[q] + 3 * (i < rcnt[q] ? i : 0)];
                                    }
                                }
                                if (rkind[q] != 3) Db[t * 32 + kk] = (__bf16)(float)v;
                            }
                        }
                        __syncthreads();
                        b8 frag[NT];
#pragma unroll
                        for (int I = 0; I < NT; ++I)
                            frag[I] = *reinterpret_cast<const b8 *>(Db + (16 * I + (lane & 15)) * 32 + 8 * (lane >> 4));
                        MfmaTiles32<NT, NW, 0>::run_bf16(frag, acc, wave, pass);
                        __syncthreads();
                    }
                    MfmaTiles32<NT, NW, 0>::store(oP, N, acc, wave, lane, pass);
                }
            }
            SLK_STAMP(15);
        }
    }
    if (tid == 0 && status) atomicOr(a.status + bidx, status);
}

// ------------------------------------------------------------------ the Msckf factor kernel
// generateSigmaPoints' Cholesky of the full covariance (Msckf.hpp:407-413, Eigen::LLT) for 32 < N <= 64, in its own
// launch: ONE wave per filter (cholp_factor: all ten tiles in registers as transposed matrix-core accumulators, the four
// pivot columns of a step factored once for all rows, no workgroup barrier, 2 KB of LDS, 116 registers -> four waves per
// SIMD, sixteen filters per CU), the packed factor handed to the step kernel through a workspace.  36.4 us per 4096
// filters (round 2, cholm_factor at 168 registers: 45.7 us; four waves per filter, two barriers per step: 68 us).
// The exact shapes of the fast path (k = 4 .. 8 clones, eight rows) and the Usckf unit-test shape run the same
// factorisation INSIDE their update kernels instead (no factor round trip through memory): this kernel serves the other
// shapes of N <= 64 and the -DSLK_MSCKF_FACTOR_KERNEL / -DSLK_USCKF_FACTOR_KERNEL builds; -DSLK_CHOL_BY_TILES: round 2's form.
#ifndef SLK_CHOL1_WAVES
#define SLK_CHOL1_WAVES 3
#endif
template <int NT, int KST = -1>
// (N <= 48: six tiles in registers -- four waves per SIMD, sixteen filters per CU: 4096 filters are one round)
__global__ __launch_bounds__(64, (NT <= 3 ? 4 : SLK_CHOL1_WAVES)) void msckf_chol_kernel(KArgs a)
{
    __shared__ __attribute__((aligned(16))) double colbuf[CholM<NT>::COLBUF];
    const int bidx = blockIdx.x, lane = threadIdx.x;
    const int N = (KST >= 0) ? 12 + 6 * KST : a.lay.N;
    const double *gP = a.P + (size_t)bidx * N * N;
    d4 acc[CholM<NT>::NTL];
#ifdef SLK_CHOL_BY_TILES
    cholm_load<NT>(acc, N, lane, [&](int i, int j) { return gP[i + (size_t)j * N]; });
    const int fail = cholm_factor<NT>(acc, a.wsL + (size_t)bidx * pk_size(N), N, colbuf, lane);
#else
    cholm_load_t<NT>(acc, N, lane, [&](int i, int j) { return gP[i + (size_t)j * N]; });
    const int fail = cholp_factor<NT>(acc, a.wsL + (size_t)bidx * pk_size(N), N, colbuf, lane);
#endif
    if (lane == 0) a.wsfail[bidx] = fail;
}

// The large states' first factorisation (N > 80, chol_blocked_mem on the global workspace) in its own launch as well: it
// needs the 16-column panel and little else in LDS (30 KB against the 90 KB of the step kernel), so several filters of a
// CU factor side by side -- at B = 512 the step kernel alone ran this latency-bound phase twice in a row on every CU.
template <int NTHREADS>
__global__ __launch_bounds__(NTHREADS, 4) void msckf_chol_big_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int flag;
    const int bidx = blockIdx.x, tid = threadIdx.x, N = a.lay.N;
    const int TN = 16 * ((N + 15) / 16);
    const double *gP = a.P + (size_t)bidx * N * N;
    const int fail = chol_blocked_mem<NTHREADS>(a.wsL + (size_t)bidx * pk_size(N), N, smem, smem + TN * 17, &flag, tid,
                                                [&](int i, int j) { return gP[i + (size_t)j * N]; });
    if (tid == 0) a.wsfail[bidx] = fail;
}
__host__ inline size_t chol_big_lds(int N) { return (size_t)(16 * ((N + 15) / 16) * 17 + 4 * 34 + 136 + 16 + 8) * sizeof(double); }

// ------------------------------------------------------------------ the Msckf predict kernel
// Msckf::predict (Msckf.hpp:89-189; state <-> clone cross-covariances stay stale, :171-182): sigma points of the current
// State's 12 x 12 block, process model, manifold mean, cov + Q.  One WAVE per filter (64-thread workgroups, 8.6 KB of
// LDS): the phase is a chain of small dependent steps, so its throughput comes from many filters per SIMD -- up to
// eight resident waves here against the single busy wave per four it had inside the fused step kernel.
// Also the Tier-B halves: emit == 1 writes the 25 sigma points, pm == SLK_MODEL_EXTERNAL takes f(X) from Yext.
#ifndef SLK_INST_UNIT     // (non-template kernels: defined in the main translation unit only)
#ifndef SLK_PRED_WAVES
#define SLK_PRED_WAVES 4     // waves per SIMD the predict kernel is compiled for
#endif
__global__ __launch_bounds__(64, SLK_PRED_WAVES) void msckf_predict_kernel(KArgs a)
{
    __shared__ __attribute__((aligned(16))) double sm[16 + 320 + 736];
    const int bidx = blockIdx.x, tid = threadIdx.x;
    const int N = a.lay.N, Nq = a.lay.Nq;
    double *mu = sm, *Lblk = sm + 16, *Pn = sm + 16 + 160, *scr = sm + 16 + 320;   // 78 + 144 + (325+300+32+72)
    double *gmean = a.mean + (size_t)bidx * Nq;
    double *gP = a.P + (size_t)bidx * N * N;
    if (tid < 13) mu[tid] = gmean[tid];
    wave_sync();
    const int st = predict_phase<false>(a, bidx, tid, [&](int i, int j) { return gP[i + (size_t)j * N]; }, Lblk, mu, Pn, scr, nullptr);
    if (st < 0) return;                              // sigma points emitted
    if (!(st & SLK_ST_LLT_FAIL)) {                   // else: predict skipped, filter unchanged
        for (int e = tid; e < 144; e += 64) gP[(e % 12) + (size_t)(e / 12) * N] = Pn[e];
        if (tid < 13) gmean[tid] = mu[tid];
    }
    if (tid == 0 && st) atomicOr(a.status + bidx, st);
}

#endif

// ------------------------------------------------------------------ MFMA fragment layout self test
#ifndef SLK_INST_UNIT
// The strict upper triangle of every covariance from its lower triangle (one workgroup per filter, 16 x 16 tiles through
// LDS: rows in, rows out).  The exact-shape update kernels store P+ as lower triangle + diagonal tiles (every kernel of this
// library READS the lower triangle only, Msckf.hpp:412, :447: Eigen::LLT; the Usckf predict keeps nothing but the lower
// triangle proper up to date); the host runs this before anything else sees the
// matrix (slk_get_state, slk_cov_device_ptr, window operations, the EKF update, ...).
__global__ __launch_bounds__(256) void slk_mirror_upper_kernel(double *P, int N)
{
    __shared__ double tile[16][17];
    double *p = P + (size_t)blockIdx.x * N * N;
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int nt = (N + 15) / 16;
    for (int I = 0; I < nt; ++I)
        for (int J = 0; J <= I; ++J) {                // lower tile (I, J) -> upper tile (J, I); a diagonal tile onto itself
            const int row = 16 * I + c, col = 16 * J + r;
            tile[r][c] = (row < N && col < N) ? p[row + (size_t)col * N] : 0.0;      // tile[col][row]
            __syncthreads();
            const int urow = 16 * J + c, ucol = 16 * I + r;                          // P(urow, ucol) = P(ucol, urow) = tile[urow - 16J][ucol - 16I]
            if (urow < N && ucol < N && urow < ucol) p[urow + (size_t)ucol * N] = tile[c][r];
            __syncthreads();
        }
}

__global__ void selftest_mfma_kernel(const double *Amat /*16x4 row-major*/, const double *Bmat /*4x16 row-major*/,
                                     double *C /*16x16 row-major*/)
{
    int l = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Amat[(l & 15) * 4 + (l >> 4)], Bmat[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

#endif

} // namespace slk

#include "slk_step_fast.hpp"
#ifndef SLK_INST_UNIT
#include "slk_general.hpp"
#endif
