// slk_ekf_tiles.hpp -- Msckf EKF update (reference src/filters/Msckf.hpp:284-349), LDS-resident tile kernel for
// m <= 128 measurement rows and N <= 64 error-state dimensions.  One 1024-thread workgroup (sixteen waves) per filter.
//
// Every matrix lives in LDS as 16 x 16 tiles (column stride 17: an MFMA operand fetch is conflict free along rows and
// along columns), and every O(n^3) step runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64):
//
//   S0 = H P H^T + R            two products; lower-triangle tiles
//   chol(S0)                    right-looking by tile columns: the 16 x 16 diagonal tile is factored AND inverted by one
//                               wave in registers (rows on lanes, pivots broadcast with v_readlane), the panel is
//                               A_IJ Dinv_J^T and the trailing update -L_IJ L_KJ^T, both MFMA
//   information blocks          (S0^-1)_bb = sum_p Li(p, r) Li(p, c) (Msckf.hpp:765-773 reads the 2 x 2 diagonal blocks
//                               only): one wave per block column of Li = L0^-1, forward substitution on tiles held in
//                               registers (an MFMA result tile is the next product's B operand as it lies), the
//                               diagonal tile of Li^T Li straight after -- Li is never stored
//   removeOutliers (:756-789)   one wave: every pair's d2 in parallel, first failing pair by ballot, the reference's
//                               two erases (the second one shifted) as lane shifts of the row index list, re-test from
//                               that pair on
//   reduceDimension (:791-816)  Householder QR with Eigen's reflector convention, panels of 16 columns: inside a panel
//                               one column per wave in registers (one barrier per reflector), compact WY
//                               (T from V^T V and tau) for the trailing columns and for thinQ = Q * I(m', N), which is
//                               formed in place over the reflectors
//   Rn = thinQ^T R' thinQ       R' gathered from global memory through the surviving-row list; R' thinQ is consumed
//                               tile row by tile row from registers
//   gain                        U = Hr P, S = U Hr^T + Rn, chol(S) as above, X = Ls^-1 U by block columns in registers;
//                               Pk - K S K^T = Pk - X^T X and K rn = X^T (Ls^-1 rn): K itself is never formed
//
// The general kernel (slk_ekf.hpp, global workspace) keeps every other shape.
#pragma once
#include "slk_ekf.hpp"

namespace slk {

constexpr int ET = 17;                 // column stride of a tile in LDS (doubles)
constexpr int ETS = 16 * ET;           // doubles per tile

__host__ __device__ inline int ekf_lt(int I, int J) { return I * (I + 1) / 2 + J; }   // lower-triangle tile (I >= J)
__host__ __device__ inline int ekf_ut(int I, int J) { return J * (J + 1) / 2 + I; }   // upper-triangle tile (I <= J)

__device__ __forceinline__ d4 ekf_mfma(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }
// lane = (c = lane & 15, g = lane >> 4); k-step ks covers k = 4 ks + g
__device__ __forceinline__ double tfA(const double *t, int ks, int c, int g) { return t[(4 * ks + g) * ET + c]; }   // element (c, 4ks+g)
__device__ __forceinline__ double tfT(const double *t, int ks, int c, int g) { return t[c * ET + 4 * ks + g]; }     // element (4ks+g, c)
__device__ __forceinline__ d4 tload(const double *t, int c, int g)
{
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = t[c * ET + g + 4 * r];
    return v;
}
__device__ __forceinline__ void tstore(double *t, int c, int g, d4 v)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) t[c * ET + g + 4 * r] = v[r];
}
__device__ __forceinline__ double bcast_lane(double x, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) { return bcast_lane(wave_inclusive_scan(x), 63); }

// Cholesky factor and inverse factor of one 16 x 16 diagonal tile, one wave: lane i (and its three copies) holds row i.
#ifdef SLK_STAMPS
#define EKF_DF_T(k) do { if (dfdbg) { long long tnow = clock64(); if (lane == 0) dfdbg[k] += tnow - tlast; tlast = tnow; } } while (0)
#else
#define EKF_DF_T(k) do { } while (0)
#endif
__device__ __forceinline__ void ekf_diag_factor(double *t, double *dinv, int lane, int row0, int *flag, long long *dfdbg = nullptr)
{
#ifdef SLK_STAMPS
    long long tlast = clock64();
#endif
    const int i = lane & 15;
    double a[16], rsv[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) a[cc] = t[cc * ET + i];
    double dg = t[i * ET + i];                 // running diagonal element of the own row: its update needs no broadcast
    int fail = -1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double d = bcast_lane(dg, j);
        if (!(d > 0.0) && fail < 0) fail = row0 + j;
        // sqrt(d) and 1 / sqrt(d) by two coupled Goldschmidt steps (two dependent operations each)
        double g0 = __builtin_amdgcn_rsq(d), h0 = 0.5 * g0;
        g0 = d * g0;
        double r0 = fma(-g0, h0, 0.5);
        g0 = fma(g0, r0, g0);
        h0 = fma(h0, r0, h0);
        r0 = fma(-g0, h0, 0.5);
        g0 = fma(g0, r0, g0);
        h0 = fma(h0, r0, h0);
        const double rs = h0 + h0;
        rsv[j] = rs;
        const double mlt = a[j] * rs;          // the multiplier of every row below the pivot
        dg = fma(-mlt, mlt, dg);
        const double l = (i > j) ? mlt : (i == j ? g0 : 0.0);
        a[j] = l;
#pragma unroll
        for (int cc = j + 1; cc < 16; ++cc) a[cc] = fma(-l, bcast_lane(mlt, cc), a[cc]);
    }
    if (lane < 16) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc)
            if (cc <= i) t[cc * ET + i] = a[cc];
        if (lane == 0 && fail >= 0 && *flag < 0) *flag = fail;
    }
    wave_sync();
    EKF_DF_T(20);
    // column i of the inverse: x_i = 1 / l_ii, x_r = -(sum_{p < r} l_rp x_p) / l_rr (l_rp: broadcast reads of the tile);
    // the terms with p < r - 1 do not wait for x_{r-1}: two dependent operations per row
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int p = 0; p + 1 < r; ++p) {
            if (p & 1) s1 = fma(t[p * ET + r], x[p], s1);
            else s0 = fma(t[p * ET + r], x[p], s0);
        }
        double sr = s0 + s1;
        if (r >= 1) sr = fma(t[(r - 1) * ET + r], x[r - 1], sr);
        x[r] = (r == i) ? rsv[r] : (r > i ? -sr * rsv[r] : 0.0);
    }
    if (lane < 16) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) dinv[i * ET + cc] = x[cc];
    }
    EKF_DF_T(21);
}

// Blocked Cholesky of the lower-triangle tiles T[ekf_lt(I, J)], NT x NT tiles (padding rows carry a unit diagonal), by
// all NW waves.  Dinv[J] = inverse of the J-th diagonal factor tile.  The factor overwrites T (strict upper parts of
// the diagonal tiles are garbage).  *flag = first non-positive pivot, or stays -1.
template <int NW>
__device__ __forceinline__ void ekf_tile_cholesky(double *T, double *Dinv, int NT, int wave, int lane, int *flag, long long *dfdbg = nullptr)
{
    const int c = lane & 15, g = lane >> 4;
    if (wave == 0) ekf_diag_factor(T, Dinv, lane, 0, flag, dfdbg);
    __syncthreads();
    for (int J = 0; J + 1 < NT; ++J) {
        const double *Dj = Dinv + J * ETS;
        for (int I = J + 1 + wave; I < NT; I += NW) {             // panel: L_IJ = A_IJ Dinv_J^T
            double *t = T + ekf_lt(I, J) * ETS;
            double af[4], bf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { af[ks] = tfA(t, ks, c, g); bf[ks] = tfA(Dj, ks, c, g); }
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = ekf_mfma(af[ks], bf[ks], acc);
            tstore(t, c, g, acc);
        }
        __syncthreads();
        const int r = NT - J - 1, cnt = r * (r + 1) / 2;
        for (int t = wave; t < cnt; t += NW) {                    // trailing tiles (I, K), J < K <= I; t = 0 is (J+1, J+1)
            int ii = 0;
            while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
            const int I = J + 1 + ii, K = J + 1 + t - ii * (ii + 1) / 2;
            double *tt = T + ekf_lt(I, K) * ETS;
            const double *li = T + ekf_lt(I, J) * ETS, *lk = T + ekf_lt(K, J) * ETS;
            d4 acc = tload(tt, c, g);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = ekf_mfma(-tfA(li, ks, c, g), tfA(lk, ks, c, g), acc);
            tstore(tt, c, g, acc);
            if (t == 0) {                                         // the next diagonal tile: factor it straight away
                wave_sync();
                ekf_diag_factor(tt, Dinv + (J + 1) * ETS, lane, 16 * (J + 1), flag, dfdbg);
            }
        }
        __syncthreads();
    }
}

// X <- L^-1 X for one block column held in registers (tile I of the column in X[I], MFMA result layout), rows I0..NT-1:
// X_I = Dinv_I (X_I - sum_{I0 <= K < I} L_IK X_K).  One wave, no barrier.
template <int MAXT>
__device__ __forceinline__ void ekf_block_forward(const double *T, const double *Dinv, int NT, int I0, d4 (&X)[MAXT], int c, int g)
{
#pragma unroll
    for (int I = 0; I < MAXT; ++I) {
        if (I >= I0 && I < NT) {
            d4 acc = X[I];
#pragma unroll
            for (int K = 0; K < I; ++K) {
                if (K >= I0) {
                    const double *l = T + ekf_lt(I, K) * ETS;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = ekf_mfma(-tfA(l, ks, c, g), X[K][ks], acc);
                }
            }
            const double *di = Dinv + I * ETS;
            d4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) r = ekf_mfma(tfA(di, ks, c, g), acc[ks], r);
            X[I] = r;
        }
    }
}

// LDS doubles of the tile kernel (dynamic part)
__host__ __device__ inline size_t ekf_tile_lds_doubles(int N, int m)
{
    const int NTR = (m + 15) / 16, NTN = (N + 15) / 16;
    const int t0 = (NTR * (NTR + 1) / 2 > NTN * NTN) ? NTR * (NTR + 1) / 2 : NTN * NTN;          // S0 tiles (P staged there first)
    const size_t gate = (size_t)(t0 + NTN * NTR) * ETS;                                          // + P H^T tiles
    const size_t qr = (size_t)(NTR * NTN + NTN * (NTN + 1) / 2 * 2 + NTN + 1 + NTN) * ETS + 256;  // Hq, Hr, Rn/S, T, G, W/Z + v
    return gate > qr ? gate : qr;
}

#ifdef SLK_STAMPS
#define EKF_DFDBG (a.dbg ? a.dbg + (size_t)blockIdx.x * 32 : nullptr)
#else
#define EKF_DFDBG nullptr
#endif
template <int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void msckf_ekf_tile_kernel(EkfArgs a)
{
    static_assert(NTHREADS == 1024, "sixteen waves: one per panel column of the Householder sweep");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NW = NTHREADS / 64;
    __shared__ int idx[136];
    __shared__ int sh[8];                  // 0 surviving rows, 2 flag S0, 3 flag S
    __shared__ double infob[3 * 64];
    __shared__ double innov[128], rq[128], tau[64], rn[64], yv[64], delta[64];
    const int tid = threadIdx.x, b = blockIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int N = a.N, Nq = a.Nq, m = a.m;
    const int NTR = (m + 15) >> 4, NTN = (N + 15) >> 4;
    double *mean = a.mean + (size_t)b * Nq, *P = a.P + (size_t)b * N * N;
    const double *z = a.z + (size_t)b * m, *zm = a.zmean + (size_t)b * m, *H = a.H + (size_t)b * m * N;
    const double *R = a.R + (size_t)b * a.r_stride;
    int status = 0;

    // ---- gate phase carve
    double *T0 = lds;                                             // lower tiles of S0 -> its factor
    double *Wt = lds + (size_t)((NTR * (NTR + 1) / 2 > NTN * NTN) ? NTR * (NTR + 1) / 2 : NTN * NTN) * ETS;   // P H^T, tile (pt, J) at J * NTN + pt
    double *Dinv = Wt;                                            // (P H^T is dead once S0 stands)
    if (tid == 0) { sh[0] = m; sh[2] = -1; sh[3] = -1; a.outliers[b] = 0u; }
    EKF_STAMP(0);
    for (int r = tid; r < m; r += NTHREADS) { innov[r] = z[r] - zm[r]; idx[r] = r; }              // :312
    if (tid < 64) tau[tid] = 0.0;
    // S0 = H P H^T + R (:765-766).  P is staged once as tiles (in the S0 region, free until the second product), every
    // fragment of H is fetched from global memory once per product and reused across the tiles of its row / column.
    double *Pt = T0;                                              // tile (I, J) of P at J * NTN + I
    for (int e = tid; e < NTN * NTN * 256; e += NTHREADS) {
        const int til = e >> 8, r = e & 15, cc = (e >> 4) & 15, i = 16 * (til % NTN) + r, j = 16 * (til / NTN) + cc;
        Pt[(size_t)til * ETS + cc * ET + r] = (i < N && j < N) ? P[i + (size_t)N * j] : 0.0;
    }
    __syncthreads();
    for (int it = wave; it < 2 * NTR; it += NW) {                 // W = P H^T: column tile J of W, two row tiles per item
        const int J = it >> 1, j = 16 * J + c;
        double bf[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int p = 4 * ks + g;
            bf[ks] = (j < m && p < N) ? H[j + (size_t)m * p] : 0.0;
        }
        for (int pt = (it & 1) * ((NTN + 1) >> 1); pt < ((it & 1) ? NTN : ((NTN + 1) >> 1)); ++pt) {
            d4 acc = {0.0, 0.0, 0.0, 0.0}, acc1 = acc;
#pragma unroll
            for (int ks = 0; ks < 16; ks += 2) {
                if (4 * ks < N) {
                    const double *ptile = Pt + (size_t)((ks >> 2) * NTN + pt) * ETS;
                    acc = ekf_mfma(tfA(ptile, ks & 3, c, g), bf[ks], acc);
                    acc1 = ekf_mfma(tfA(ptile, (ks + 1) & 3, c, g), bf[ks + 1], acc1);
                }
            }
            tstore(Wt + (size_t)(J * NTN + pt) * ETS, c, g, acc + acc1);
        }
    }
    __syncthreads();
    for (int it = wave; it < 2 * NTR; it += NW) {                 // S0 = H W + R: row I of lower tiles, split in two halves
        const int I = it >> 1, i = 16 * I + c, nh = (I + 2) >> 1;
        double af[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int p = 4 * ks + g;
            af[ks] = (i < m && p < N) ? H[i + (size_t)m * p] : 0.0;
        }
        for (int J = (it & 1) ? nh : 0; J < ((it & 1) ? I + 1 : nh); ++J) {
            d4 acc = {0.0, 0.0, 0.0, 0.0}, acc1 = acc;
#pragma unroll
            for (int ks = 0; ks < 16; ks += 2) {
                if (4 * ks < N) {
                    const double *wt = Wt + (size_t)(J * NTN + (ks >> 2)) * ETS;
                    acc = ekf_mfma(af[ks], tfT(wt, ks & 3, c, g), acc);
                    acc1 = ekf_mfma(af[ks + 1], tfT(wt, (ks + 1) & 3, c, g), acc1);
                }
            }
            acc = acc + acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + g + 4 * r, col = 16 * J + c;
                acc[r] = (row < m && col < m) ? acc[r] + R[row + (size_t)m * col] : (row == col ? 1.0 : 0.0);
            }
            tstore(T0 + (size_t)ekf_lt(I, J) * ETS, c, g, acc);
        }
    }
    __syncthreads();
    EKF_STAMP(1);
    ekf_tile_cholesky<NW>(T0, Dinv, NTR, wave, lane, &sh[2], EKF_DFDBG);
    EKF_STAMP(2);
    if (sh[2] >= 0) {
        status |= SLK_ST_SINGULAR;              // the reference would invert an indefinite matrix with PartialPivLU
    } else {
        // ---- information blocks: block column J of Li = L0^-1 in registers, then the diagonal tile of Li^T Li
        if (wave < NTR) {
            const int J = wave;
            d4 X[8];
#pragma unroll
            for (int I = 0; I < 8; ++I)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[I][r] = (I == J && g + 4 * r == c) ? 1.0 : 0.0;
            ekf_block_forward<8>(T0, Dinv, NTR, J, X, c, g);
            d4 G = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int K = 0; K < 8; ++K)
                if (K >= J && K < NTR)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) G = ekf_mfma(X[K][ks], X[K][ks], G);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = g + 4 * r, blk = 8 * J + (row >> 1);
                if (16 * J + row < m) {
                    if (row == c) infob[3 * blk + ((row & 1) ? 2 : 0)] = G[r];
                    else if (!(row & 1) && c == row + 1) infob[3 * blk + 1] = G[r];
                }
            }
        }
        __syncthreads();
        EKF_STAMP(3);
        // ---- removeOutliers (:767-787): lane i tests pair i of the current row list with block i of the information
        if (wave == 0) {
            int cnt = m, start = 0;
            unsigned nout = 0;
            for (;;) {
                const int i = lane;
                const bool valid = i >= start && i < cnt / 2;
                bool bad = false;
                if (valid && a.gate) {
                    const double r0 = innov[idx[2 * i]], r1 = innov[idx[2 * i + 1]];
                    const double i00 = infob[3 * i], i01 = infob[3 * i + 1], i11 = infob[3 * i + 2];
                    const double d2 = r0 * (i00 * r0 + i01 * r1) + r1 * (i01 * r0 + i11 * r1);
                    bad = !(d2 < 5.99);                                // chi2_0.95(2), Msckf.hpp:861-865
                }
                const unsigned long long mask = __ballot(bad);
                if (mask == 0ull) break;
                const int f = __ffsll((long long)mask) - 1;
                for (int rep = 0; rep < 2; ++rep) {                    // removeRow semantics, :688-697
                    const int pos = 2 * f + rep, numRows = cnt - 1;
                    const int q0 = lane, q1 = lane + 64;
                    const int v0 = (q0 >= pos && q0 < numRows) ? idx[q0 + 1] : 0;
                    const int v1 = (q1 >= pos && q1 < numRows) ? idx[q1 + 1] : 0;
                    wave_sync();
                    if (q0 >= pos && q0 < numRows) idx[q0] = v0;
                    if (q1 >= pos && q1 < numRows) idx[q1] = v1;
                    wave_sync();
                    cnt = numRows;
                }
                nout++;
                start = f;
            }
            if (lane == 0) { sh[0] = cnt; a.outliers[b] = nout; }
        }
        __syncthreads();
        const int mm = sh[0];
        EKF_STAMP(4);
        if (mm > 0 && mm < N) {
            status |= SLK_ST_EKF_ROWS;          // reduceDimension would read R.block(0,0,N,N) out of range (:806)
        } else if (mm > 0) {
            // ---- QR phase carve
            const int NTr = (mm + 15) >> 4;                                        // row tiles in use
            double *Hq = lds;                                                       // tile (I, t) at t * NTR + I
            double *HrT = Hq + (size_t)NTR * NTN * ETS;                              // upper tiles of Hr
            double *RnT = HrT + (size_t)(NTN * (NTN + 1) / 2) * ETS;                 // lower tiles of Rn -> S -> its factor
            double *Tp = RnT + (size_t)(NTN * (NTN + 1) / 2) * ETS;                  // T of every panel
            double *Gt = Tp + (size_t)NTN * ETS;
            double *Wz = Gt + ETS;
            double *vbuf = Wz + (size_t)NTN * ETS;
            auto hq = [&](int I, int t) -> double * { return Hq + (size_t)(t * NTR + I) * ETS; };
            for (int e = tid; e < NTR * NTN * 256; e += NTHREADS) {                  // gated rows of H, zero padded
                const int til = e >> 8, r = e & 15, cc = (e >> 4) & 15, I = til % NTR, t = til / NTR;
                const int i = 16 * I + r, j = 16 * t + cc;
                Hq[(size_t)til * ETS + cc * ET + r] = (i < mm && j < N) ? H[idx[i] + (size_t)m * j] : 0.0;
            }
            for (int i = tid; i < 128; i += NTHREADS) rq[i] = (i < mm) ? innov[idx[i]] : 0.0;
            __syncthreads();
            EKF_STAMP(5);
            // reflector element (i, j) of the panel stored in tile (I, p): unit diagonal, zero above
            auto vT = [&](const double *t, int I, int p, int ks) -> double {        // A operand [a][i]: element (4ks+g, c)
                const int i = 16 * I + 4 * ks + g, j = 16 * p + c;
                const double v = tfT(t, ks, c, g);
                return (i > j) ? v : (i == j ? 1.0 : 0.0);
            };
            auto vA = [&](const double *t, int I, int p, int ks) -> double {        // A operand [i][a]: element (c, 4ks+g)
                const int i = 16 * I + c, j = 16 * p + 4 * ks + g;
                const double v = tfA(t, ks, c, g);
                return (i > j) ? v : (i == j ? 1.0 : 0.0);
            };
            // V_p^T [V_p | column tiles p+1..] (sweep) or V_p^T [Q column tiles p+1..] (thinQ): tile t of the product on
            // wave t, its row-tile range dealt to four waves (t, t + 4, t + 8, t + 12); the partial tiles meet in LDS
            // and are added in a fixed order.  Returns the finished tile to waves 0..3.
            auto wy_products = [&](int p, bool with_g, double *Sc) -> d4 {
                const int ntrail = NTN - 1 - p, t = wave & 3, part = wave >> 2;
                const int nt = with_g ? ntrail + 1 : ntrail, ts = with_g ? t : t - 1;      // tiles per part, slot of tile t
                const int np = (NTN == 1) ? 1 : 4;                                          // (N <= 16: no room for partial tiles)
                const bool active = t <= ntrail && (with_g || t >= 1) && part < np;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                if (active) {
                    for (int I = p + part; I < NTr; I += np) {
                        const double *vt = hq(I, p), *bt = hq(I, p + t);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const double av = vT(vt, I, p, ks);
                            acc = ekf_mfma(av, (t == 0) ? av : tfT(bt, ks, c, g), acc);
                        }
                    }
                    if (part) tstore(Sc + (size_t)((part - 1) * nt + ts) * ETS, c, g, acc);
                }
                __syncthreads();
                if (active && part == 0)
                    for (int q = 1; q < np; ++q) acc = acc + tload(Sc + (size_t)((q - 1) * nt + ts) * ETS, c, g);
                return acc;
            };
#ifdef SLK_STAMPS
            long long tsw = clock64();
#define EKF_SW_T(k) do { if (a.dbg && tid == 0) { long long tn = clock64(); a.dbg[(size_t)blockIdx.x * 32 + (k)] += tn - tsw; tsw = tn; } } while (0)
#else
#define EKF_SW_T(k) do { } while (0)
#endif
            for (int p = 0; p < NTN; ++p) {
                const int c0 = 16 * p, nb = (N - c0 < 16) ? N - c0 : 16;
                const bool own = wave < nb;
                const int col = c0 + wave;
                double x0 = 0.0, x1 = 0.0;
                if (own) {
                    x0 = (lane < 16 * NTR) ? hq(lane >> 4, p)[wave * ET + (lane & 15)] : 0.0;
                    x1 = (lane + 64 < 16 * NTR) ? hq((lane + 64) >> 4, p)[wave * ET + (lane & 15)] : 0.0;
                }
                for (int jj = 0; jj < nb; ++jj) {
                    const int kk = c0 + jj;
                    double *vb = vbuf + (jj & 1) * 128;
                    if (wave == jj) {                                  // Eigen makeHouseholder
                        const double tail = wave_sum(((lane > kk) ? x0 * x0 : 0.0) + x1 * x1);
                        const double cc0 = bcast_lane(x0, kk);
                        double beta, tk, rden;
                        if (tail <= 2.2250738585072014e-308) { tk = 0.0; beta = cc0; rden = 0.0; }
                        else {
                            // beta = -sign(c0) |x|, tau = (beta - c0) / beta = 1 + |c0| / |x|, 1 / (c0 - beta) =
                            // sign(c0) / (|c0| + |x|): one reciprocal square root and one reciprocal, Newton-refined
                            double nrm, rnrm;
                            rsqrt_pivot(cc0 * cc0 + tail, nrm, rnrm);
                            const double ac = fabs(cc0), sg = (cc0 >= 0.0) ? 1.0 : -1.0, dd = ac + nrm;
                            double rr = __builtin_amdgcn_rcp(dd);
                            rr = rr * fma(-dd, rr, 2.0);
                            rr = rr * fma(-dd, rr, 2.0);
                            beta = -sg * nrm;
                            tk = fma(ac, rnrm, 1.0);
                            rden = sg * rr;
                        }
                        const double v0 = (lane > kk) ? x0 * rden : (lane == kk ? 1.0 : 0.0), v1 = x1 * rden;
                        vb[lane] = v0;
                        vb[lane + 64] = v1;
                        if (lane == 0) tau[kk] = tk;
                        x0 = (lane > kk) ? v0 : (lane == kk ? beta : x0);
                        x1 = v1;
                        if (lane < 16 * NTR) hq(lane >> 4, p)[wave * ET + (lane & 15)] = x0;
                        if (lane + 64 < 16 * NTR) hq((lane + 64) >> 4, p)[wave * ET + (lane & 15)] = x1;
                    }
                    __syncthreads();
                    if (own && wave > jj) {
                        const double v0 = vb[lane], v1 = vb[lane + 64];
                        const double w = tau[kk] * wave_sum(v0 * x0 + v1 * x1);
                        x0 = fma(-v0, w, x0);
                        x1 = fma(-v1, w, x1);
                    }
                }
                (void)col;
                EKF_SW_T(22);
                // ---- compact WY of the panel: G = V^T V (wave 0), W_t = V^T A_t for the trailing column tiles (waves 1..)
                const int ntrail = NTN - 1 - p;
                const d4 wacc = wy_products(p, true, HrT);                 // (Hr / Rn regions are free during the sweep)
                EKF_SW_T(23);
                if (wave == 0) {
                    tstore(Gt, c, g, wacc);
                    wave_sync();
                    // T (upper triangular): column a by back substitution on T^-1 = striu(G) + diag(1 / tau); the term
                    // with the element found last is added last (two dependent operations per element)
                    const int aa = lane & 15;
                    double t[16];
#pragma unroll
                    for (int i = 15; i >= 0; --i) {
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int q = i + 2; q < 16; ++q) {
                            if (q & 1) s1 = fma(Gt[q * ET + i], t[q], s1);
                            else s0 = fma(Gt[q * ET + i], t[q], s0);
                        }
                        double sr = s0 + s1;
                        if (i + 1 < 16) sr = fma(Gt[(i + 1) * ET + i], t[i + 1], sr);
                        const double ti = tau[c0 + i];
                        t[i] = (i == aa) ? ti : (i < aa ? -ti * sr : 0.0);
                    }
                    if (lane < 16) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) Tp[(size_t)p * ETS + aa * ET + i] = t[i];
                    }
                }
                __syncthreads();
                EKF_SW_T(24);
                if (wave >= 1 && wave <= ntrail && wave < 4) {             // Z_t = T^T W_t
                    const double *tp = Tp + (size_t)p * ETS;
                    d4 zz = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) zz = ekf_mfma(tfT(tp, ks, c, g), wacc[ks], zz);
                    tstore(Wz + (size_t)(wave - 1) * ETS, c, g, zz);
                }
                __syncthreads();
                for (int t = wave; t < ntrail * (NTr - p); t += NW) {      // A_(I, t) -= V_I Z_t
                    const int I = p + t % (NTr - p), tt = t / (NTr - p);
                    double *at = hq(I, p + 1 + tt);
                    const double *vt = hq(I, p), *zt = Wz + (size_t)tt * ETS;
                    d4 acc = tload(at, c, g);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = ekf_mfma(-vA(vt, I, p, ks), tfT(zt, ks, c, g), acc);
                    tstore(at, c, g, acc);
                }
                __syncthreads();
                EKF_SW_T(25);
            }
            EKF_STAMP(6);
            // ---- Hr = R factor (upper N x N), then thinQ in place over the reflectors
            for (int e = tid; e < NTN * (NTN + 1) / 2 * 256; e += NTHREADS) {
                const int til = e >> 8, r = e & 15, cc = (e >> 4) & 15;
                int J = 0;
                while ((J + 1) * (J + 2) / 2 <= til) ++J;
                const int I = til - J * (J + 1) / 2;
                const int i = 16 * I + r, j = 16 * J + cc;
                HrT[(size_t)til * ETS + cc * ET + r] = (i <= j && j < N) ? hq(I, J)[cc * ET + r] : 0.0;
            }
            __syncthreads();
            for (int e = tid; e < NTN * (NTN - 1) / 2 * 256; e += NTHREADS) {       // strict upper tiles of Q start at zero
                const int til = e >> 8, r = e & 15, cc = (e >> 4) & 15;
                int J = 1;
                while (J * (J + 1) / 2 <= til) ++J;
                const int I = til - J * (J - 1) / 2;
                hq(I, J)[cc * ET + r] = 0.0;
            }
            __syncthreads();
            for (int p = NTN - 1; p >= 0; --p) {
                const int c0 = 16 * p;
                const int ntrail = NTN - 1 - p;
                const double *tp = Tp + (size_t)p * ETS;
                d4 wacc = wy_products(p, false, RnT);                       // W_t = V_p^T Q_t (the Rn region is still free)
                if (wave <= ntrail && wave < 4) {                           // Z_t = T W_t
                    if (wave == 0) {
                        const double *dt = hq(p, p);                        // V^T E_p = (diagonal tile of V)^T
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int aa = g + 4 * r;
                            wacc[r] = (c0 + c < N) ? ((c > aa) ? dt[aa * ET + c] : (c == aa ? 1.0 : 0.0)) : 0.0;
                        }
                    }
                    d4 zz = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) zz = ekf_mfma(tfA(tp, ks, c, g), wacc[ks], zz);
                    tstore(Wz + (size_t)wave * ETS, c, g, zz);
                }
                __syncthreads();
                for (int t = wave; t < ntrail * (NTr - p); t += NW) {      // Q_(I, t) -= V_I Z_t, t > p
                    const int I = p + t % (NTr - p), tt = 1 + t / (NTr - p);
                    double *at = hq(I, p + tt);
                    const double *vt = hq(I, p), *zt = Wz + (size_t)tt * ETS;
                    d4 acc = tload(at, c, g);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = ekf_mfma(-vA(vt, I, p, ks), tfT(zt, ks, c, g), acc);
                    tstore(at, c, g, acc);
                }
                __syncthreads();
                for (int I = p + wave; I < NTr; I += NW) {                 // the panel's own columns: E_p - V_I Z_p
                    double *vt = hq(I, p);
                    double af[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) af[ks] = -vA(vt, I, p, ks);
                    d4 acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = (I == p && g + 4 * r == c && c0 + c < N) ? 1.0 : 0.0;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = ekf_mfma(af[ks], tfT(Wz, ks, c, g), acc);
                    tstore(vt, c, g, acc);
                }
                __syncthreads();
            }
            EKF_STAMP(7);
            // ---- rn = thinQ^T innovation (:808), Rn = thinQ^T R' thinQ (:806-812), lower tiles
            for (int j = tid >> 4; j < N; j += NTHREADS / 16) {
                const int part = tid & 15;
                const double *qt = Hq + (size_t)((j >> 4) * NTR) * ETS + (j & 15) * ET;
                double s = 0.0;
                for (int I = 0; I < NTr; ++I) s = fma(qt[(size_t)I * ETS + part], rq[16 * I + part], s);
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                s += __shfl_xor(s, 8, 64);
                if (part == 0) rn[j] = s;
            }
            {
                const int bt = wave & 3, q = wave >> 2;
                d4 racc[4];
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) racc[aa] = d4{0.0, 0.0, 0.0, 0.0};
                if (bt < NTN) {
                    for (int I = 2 * q; I < 2 * q + 2 && I < NTr; ++I) {
                        const int ri = (16 * I + c < mm) ? idx[16 * I + c] : -1;
                        d4 t1 = {0.0, 0.0, 0.0, 0.0}, t1b = t1;
                        for (int K0 = 0; K0 < NTr; K0 += 4) {                  // four tile columns of R' in flight at a time
                            double af[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const int pp = 16 * K0 + 4 * u + g;               // (u = 4 (K - K0) + ks)
                                af[u] = (ri >= 0 && pp < mm) ? R[ri + (size_t)m * idx[pp]] : 0.0;
                            }
#pragma unroll
                            for (int kq = 0; kq < 4; ++kq) {
                                if (K0 + kq < NTr) {
                                    const double *qk = hq(K0 + kq, bt);
#pragma unroll
                                    for (int ks = 0; ks < 4; ks += 2) {
                                        t1 = ekf_mfma(af[4 * kq + ks], tfT(qk, ks, c, g), t1);
                                        t1b = ekf_mfma(af[4 * kq + ks + 1], tfT(qk, ks + 1, c, g), t1b);
                                    }
                                }
                            }
                        }
                        t1 = t1 + t1b;
#pragma unroll
                        for (int aa = 0; aa < 4; ++aa) {
                            if (aa >= bt && aa < NTN) {
                                const double *qa = hq(I, aa);
#pragma unroll
                                for (int ks = 0; ks < 4; ++ks) racc[aa] = ekf_mfma(tfT(qa, ks, c, g), t1[ks], racc[aa]);
                            }
                        }
                    }
                }
                for (int qq = 0; qq < 4; ++qq) {                           // the four row groups add up in a fixed order
                    if (q == qq && bt < NTN) {
#pragma unroll
                        for (int aa = 0; aa < 4; ++aa) {
                            if (aa >= bt && aa < NTN) {
                                double *rt = RnT + (size_t)ekf_lt(aa, bt) * ETS;
                                d4 v = racc[aa];
                                if (qq > 0) v = v + tload(rt, c, g);
                                tstore(rt, c, g, v);
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            EKF_STAMP(8);
            // ---- gain phase: U = Hr P (= T2^T, :324-325), S = U Hr^T + Rn; thinQ (Hq) is dead: U and Dinv take its place
            double *UT = Hq;                                               // tile (I, J) at J * NTN + I
            double *Dinv2 = Tp;                                            // (the panels' T matrices are dead)
            for (int t = wave; t < NTN * NTN; t += NW) {
                const int I = t % NTN, J = t / NTN;
                const int j = 16 * J + c;
                d4 acc = {0.0, 0.0, 0.0, 0.0}, acc1 = acc;
                for (int Pt = I; Pt < NTN; ++Pt) {
                    const double *ht = HrT + (size_t)ekf_ut(I, Pt) * ETS;
                    double bf[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int pp = 16 * Pt + 4 * ks + g;
                        bf[ks] = (j < N && pp < N) ? P[j + (size_t)N * pp] : 0.0;       // P(pp, j) = P(j, pp)
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ks += 2) {
                        acc = ekf_mfma(tfA(ht, ks, c, g), bf[ks], acc);
                        acc1 = ekf_mfma(tfA(ht, ks + 1, c, g), bf[ks + 1], acc1);
                    }
                }
                tstore(UT + (size_t)t * ETS, c, g, acc + acc1);
            }
            __syncthreads();
            for (int t = wave; t < NTN * (NTN + 1) / 2; t += NW) {
                int I = 0;
                while ((I + 1) * (I + 2) / 2 <= t) ++I;
                const int J = t - I * (I + 1) / 2;
                double *st = RnT + (size_t)t * ETS;
                d4 acc = tload(st, c, g);
                for (int Pt = J; Pt < NTN; ++Pt) {
                    const double *ut = UT + (size_t)(Pt * NTN + I) * ETS, *ht = HrT + (size_t)ekf_ut(J, Pt) * ETS;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = ekf_mfma(tfA(ut, ks, c, g), tfA(ht, ks, c, g), acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * I + g + 4 * r, col = 16 * J + c;
                    if (row >= N || col >= N) acc[r] = (row == col) ? 1.0 : 0.0;
                }
                tstore(st, c, g, acc);
            }
            __syncthreads();
            EKF_STAMP(9);
            ekf_tile_cholesky<NW>(RnT, Dinv2, NTN, wave, lane, &sh[3], EKF_DFDBG);
            EKF_STAMP(10);
            if (sh[3] >= 0) {
                status |= SLK_ST_SINGULAR;
            } else {
                // X = Ls^-1 U by block columns (wave = block column), y = Ls^-1 rn (wave NTN)
                if (wave <= NTN) {
                    d4 X[4];
#pragma unroll
                    for (int I = 0; I < 4; ++I) {
                        if (wave < NTN) {
                            if (I < NTN) X[I] = tload(UT + (size_t)(wave * NTN + I) * ETS, c, g);
                            else X[I] = d4{0.0, 0.0, 0.0, 0.0};
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) X[I][r] = (c == 0 && 16 * I + g + 4 * r < N) ? rn[16 * I + g + 4 * r] : 0.0;
                        }
                    }
                    ekf_block_forward<4>(RnT, Dinv2, NTN, 0, X, c, g);
#pragma unroll
                    for (int I = 0; I < 4; ++I) {
                        if (I < NTN) {
                            if (wave < NTN) tstore(UT + (size_t)(wave * NTN + I) * ETS, c, g, X[I]);
                            else if (c == 0) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) yv[16 * I + g + 4 * r] = X[I][r];
                            }
                        }
                    }
                }
                __syncthreads();
                EKF_STAMP(11);
                // Pk - K S K^T = Pk - X^T X (:330), delta = K rn = X^T y
                if (tid < N) {
                    const double *xt = UT + (size_t)((tid >> 4) * NTN) * ETS + (tid & 15) * ET;
                    double s = 0.0;
                    for (int k = 0; k < 16 * NTN; ++k) s = fma(xt[(size_t)(k >> 4) * ETS + (k & 15)], yv[k], s);
                    delta[tid] = s;
                }
                for (int t = wave; t < NTN * (NTN + 1) / 2; t += NW) {
                    int A = 0;
                    while ((A + 1) * (A + 2) / 2 <= t) ++A;
                    const int Bc = t - A * (A + 1) / 2;
                    d4 acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * A + g + 4 * r, col = 16 * Bc + c;
                        acc[r] = (row < N && col < N) ? P[row + (size_t)N * col] : 0.0;
                    }
                    for (int I = 0; I < NTN; ++I) {
                        const double *xa = UT + (size_t)(A * NTN + I) * ETS, *xb = UT + (size_t)(Bc * NTN + I) * ETS;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) acc = ekf_mfma(-tfT(xa, ks, c, g), tfT(xb, ks, c, g), acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * A + g + 4 * r, col = 16 * Bc + c;
                        if (row < N && col < N) {
                            if (A != Bc || row >= col) {
                                P[row + (size_t)N * col] = acc[r];
                                if (row != col) P[col + (size_t)N * row] = acc[r];
                            }
                        }
                    }
                }
                __syncthreads();
                // mu <- mu [+] delta (:331; MultiState boxplus, State.hpp:418-434)
                for (int blk = tid; blk <= a.k; blk += NTHREADS) {
                    const int to = blk ? 12 + 6 * (blk - 1) : 0, so = blk ? 13 + 7 * (blk - 1) : 0;
                    for (int cc = 0; cc < 3; ++cc) mean[so + cc] += delta[to + cc];
                    stq(mean + so + 3, qmul(ldq(mean + so + 3), so3_exp(delta[to + 3], delta[to + 4], delta[to + 5])));
                    if (blk == 0) for (int cc = 0; cc < 6; ++cc) mean[7 + cc] += delta[6 + cc];
                }
            }
        }
    }
    EKF_STAMP(12);
    if (tid == 0 && status) atomicOr(a.status + b, status);
}

} // namespace slk
