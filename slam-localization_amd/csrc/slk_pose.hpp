// slk_pose.hpp -- batched pose-with-uncertainty ops next to the filter hot path:
//   * TransformWithUncertainty::operator*  (reference src/core/Transform.cpp:215-254, Jacobians :35-137:
//     Pennec & Thirion, "A framework for uncertainty and validation of 3-D registration methods", IJCV 1997)
//   * DeadReckon::updatePose, Affine3d overload (src/core/DeadReckon.hpp:306-330) and RigidBodyState overload
//     (:129-239) with both its branches (useTranforWithUncertainty on / off)
//   * AdaptiveAttitudeCov::matrix (src/filters/MeasurementModels.hpp:181-286), whose result is the R of update()
// One thread per filter: a few hundred flops on a 6x6 / 3x3 problem each, no reuse between filters.  A transform is
// pos[3] quat[4: x,y,z,w]; 6x6 covariances are column-major in the reference's [r t] order; 3x3 temporaries are
// row-major m[3*i + j].
#pragma once
#include <hip/hip_runtime.h>

#include "slk_math.hpp"

namespace slk {

// Eigen QuaternionBase::toRotationMatrix (the reference goes through Eigen::Affine3d)
__device__ __forceinline__ void quat_to_rot(const double *q, double *R)
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen::Quaterniond(rotation matrix) (Transform.cpp:222-225 takes the quaternions from linear())
__device__ __forceinline__ void rot_to_quat(const double *m, double *q)
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[3 * i + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[3 * i + i] - m[3 * j + j] - m[3 * k + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}

// q_to_r (Transform.cpp:44-48): Eigen::AngleAxisd(q), angle = 2 atan2(|vec|, |w|), axis = vec / (+-|vec|)
__device__ __forceinline__ void q_to_r(const double *q, double *r)
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n != 0) {
        const double angle = 2 * atan2(n, fabs(q[3]));
        if (q[3] < 0) n = -n;
        r[0] = q[0] / n * angle; r[1] = q[1] / n * angle; r[2] = q[2] / n * angle;
    } else { r[0] = r[1] = r[2] = 0; }
}

__device__ __forceinline__ void skew3(const double *r, double *S)
{
    S[0] = 0; S[1] = -r[2]; S[2] = r[1];
    S[3] = r[2]; S[4] = 0; S[5] = -r[0];
    S[6] = -r[1]; S[7] = r[0]; S[8] = 0;
}

template <int N, int K, int M>
__device__ __forceinline__ void mmul(const double *A, const double *B, double *C)      // row-major (N x K)(K x M)
{
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double s = 0;
#pragma unroll
            for (int p = 0; p < K; ++p) s += A[i * K + p] * B[p * M + j];
            C[i * M + j] = s;
        }
}

// dq_by_dr (Transform.cpp:64-76), 4 x 3, quaternion ordered (w, x, y, z)
__device__ __forceinline__ void dq_by_dr(const double *q, double *D)
{
    double r[3];
    q_to_r(q, r);
    const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    const double theta = sqrt(th2);
    const double kappa = 0.5 - theta * theta / 48.0, lambda = 1.0 / 24.0 * (1.0 - theta * theta / 40.0);
    for (int j = 0; j < 3; ++j) D[j] = -q[j] / 2.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[3 * (i + 1) + j] = kappa * (i == j) - lambda * r[i] * r[j];
}

// dr_by_dq (Transform.cpp:78-89), 3 x 4
__device__ __forceinline__ void dr_by_dq(const double *q, double *D)
{
    const double mu = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const double sg = q[3] > 0 ? 1.0 : -1.0;
    const double tau = 2.0 * sg * (1.0 + mu * mu / 6.0), nu = -2.0 * sg * (2.0 / 3.0 + mu * mu / 5.0);
    for (int i = 0; i < 3; ++i) {
        D[4 * i] = -2 * q[i];
        for (int j = 0; j < 3; ++j) D[4 * i + 1 + j] = tau * (i == j) + nu * q[i] * q[j];
    }
}

// dq2q1_by_dq1(q2): sgn = +1, dq2q1_by_dq2(q1): sgn = -1 (Transform.cpp:91-105), 4 x 4
__device__ __forceinline__ void dq2q1_by(const double *q, double sgn, double *M)
{
    double S[9];
    skew3(q, S);
    for (int i = 0; i < 16; ++i) M[i] = 0;
    for (int j = 0; j < 3; ++j) { M[1 + j] = -q[j]; M[4 * (1 + j)] = q[j]; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[4 * (1 + i) + 1 + j] = sgn * S[3 * i + j];
    for (int i = 0; i < 4; ++i) M[5 * i] += q[3];
}

// dr2r1_by_r1 / dr2r1_by_r2 (Transform.cpp:107-121)
__device__ __forceinline__ void dr2r1_by(const double *q, const double *qa, double sgn, const double *qb, double *J)
{
    double A[12], M[16], B[12], T[12];
    dr_by_dq(q, A);
    dq2q1_by(qa, sgn, M);
    dq_by_dr(qb, B);
    mmul<3, 4, 4>(A, M, T);
    mmul<3, 4, 3>(T, B, J);
}

// drx_by_dr (Transform.cpp:123-137)
__device__ __forceinline__ void drx_by_dr(const double *q, const double *x, double *J)
{
    double r[3];
    q_to_r(q, r);
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const double alpha = 1.0 - theta * theta / 6.0, beta = 0.5 - theta * theta / 24.0;
    const double gamma = 1.0 / 3.0 - theta * theta / 30.0, delta = -1.0 / 12.0 + theta * theta / 180.0;
    double Sx[9], Sr[9], A[9], B[9], T1[9], T2[9], T3[9];
    skew3(x, Sx);
    skew3(r, Sr);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[3 * i + j] = gamma * r[i] * r[j] - beta * Sr[3 * i + j] + alpha * (i == j);
            B[3 * i + j] = delta * r[i] * r[j] + 2.0 * beta * (i == j);
        }
    mmul<3, 3, 3>(Sx, A, T1);
    mmul<3, 3, 3>(Sr, Sx, T2);
    mmul<3, 3, 3>(T2, B, T3);
    for (int i = 0; i < 9; ++i) J[i] = -T1[i] - T3[i];
}

// cov (6x6 column-major) += J C J^T, J = [[J00, 0], [J10, J11]] with row-major 3x3 blocks (J01 is zero in both uses)
__device__ __forceinline__ void add_jcjt(const double *J00, const double *J10, const double *J11, const double *C, double *cov)
{
    double J[36], T[36];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            J[6 * i + j] = J00[3 * i + j]; J[6 * i + 3 + j] = 0.0;
            J[6 * (3 + i) + j] = J10 ? J10[3 * i + j] : 0.0; J[6 * (3 + i) + 3 + j] = J11[3 * i + j];
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int p = 0; p < 6; ++p) s += J[6 * i + p] * C[p + 6 * j];
            T[6 * i + j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int p = 0; p < 6; ++p) s += T[6 * i + p] * J[6 * j + p];
            cov[i + 6 * j] += s;
        }
}

// TransformWithUncertainty::operator* (Transform.cpp:215-254): out = t2 * t1; cov1 / cov2 null = no uncertainty
__device__ inline void transform_compose(const double *t2, const double *cov2, const double *t1, const double *cov1,
                                         double *out_t, double *out_cov)
{
    double R1[9], R2[9], R[9], q1[4], q2[4];
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    quat_to_rot(t1 + 3, R1);
    quat_to_rot(t2 + 3, R2);
    mmul<3, 3, 3>(R2, R1, R);
    for (int i = 0; i < 3; ++i) out_t[i] = R2[3 * i] * t1[0] + R2[3 * i + 1] * t1[1] + R2[3 * i + 2] * t1[2] + t2[i];
    rot_to_quat(R, out_t + 3);
    for (int i = 0; i < 36; ++i) out_cov[i] = 0;
    if (!cov1 && !cov2) return;                                                     // :219-220
    rot_to_quat(R1, q1);                                                            // :222-225
    rot_to_quat(R2, q2);
    double q[4];
    stq(q, qmul(ldq(q2), ldq(q1)));
    if (cov1) {                                                                     // :232-239
        double J00[9];
        dr2r1_by(q, q2, 1.0, q1, J00);
        add_jcjt(J00, nullptr, R2, cov1, out_cov);
    }
    if (cov2) {                                                                     // :241-248
        double J00[9], J10[9];
        dr2r1_by(q, q1, -1.0, q2, J00);
        drx_by_dr(q2, t1, J10);
        add_jcjt(J00, J10, I3, cov2, out_cov);
    }
}

// ---- batch kernels: one thread per filter ------------------------------------------------------------------------
// t2 / t1 [B][7], cov2 / cov1 [B][36] or null, additive = DeadReckon.hpp:317-323 (post = prev * delta, cov = prev + delta)
// (one thread per filter with 6 x 6 Jacobian chains in registers: 64-thread workgroups at one wave per SIMD give the
// compiler the whole register file instead of scratch memory; the op is a few hundred flops per filter, latency-bound)
__global__ __launch_bounds__(64, 1) void transform_compose_kernel(int B, const double *t2, const double *cov2, const double *t1, const double *cov1,
                                         double *t_out, double *cov_out, int additive)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double a2[7], a1[7], c2[36], c1[36], to[7], co[36];
    for (int i = 0; i < 7; ++i) { a2[i] = t2[(size_t)b * 7 + i]; a1[i] = t1[(size_t)b * 7 + i]; }
    if (cov2) for (int i = 0; i < 36; ++i) c2[i] = cov2[(size_t)b * 36 + i];
    if (cov1) for (int i = 0; i < 36; ++i) c1[i] = cov1[(size_t)b * 36 + i];
    if (additive) {
        transform_compose(a2, nullptr, a1, nullptr, to, co);
        for (int i = 0; i < 36; ++i) co[i] = (cov2 ? c2[i] : 0.0) + (cov1 ? c1[i] : 0.0);
    } else {
        transform_compose(a2, cov2 ? c2 : nullptr, a1, cov1 ? c1 : nullptr, to, co);
    }
    for (int i = 0; i < 7; ++i) t_out[(size_t)b * 7 + i] = to[i];
    if (cov_out) for (int i = 0; i < 36; ++i) cov_out[(size_t)b * 36 + i] = co[i];
}

// DeadReckon::updatePose, RigidBodyState overload (DeadReckon.hpp:129-239).  Records (include/slk.h):
//   prev [25] = pos quat cov_position[9] cov_orientation[9]; post [49] = that + velocity cov_velocity angular_velocity
//   cov_angular_velocity (in/out); delta [31] = pose record + velocity angular_velocity
__global__ __launch_bounds__(64, 1) void dead_reckon_pose_kernel(int B, const double *u, int u_stride, const double *velcov, int c_stride,
                                        const double *prev, double *post, double *delta, int use_tf)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double uu[13], vc[36], pv[25], d13[13], dl[31];
    const double *us = u + (size_t)b * u_stride, *cs = velcov + (size_t)b * c_stride;
    for (int i = 0; i < 13; ++i) uu[i] = us[i];
    for (int i = 0; i < 36; ++i) vc[i] = cs[i];
    for (int i = 0; i < 25; ++i) pv[i] = prev[(size_t)b * 25 + i];
    double *po = post + (size_t)b * 49;
    const double dt = uu[0];
    dead_reckon_delta(uu, d13);
    for (int i = 0; i < 7; ++i) dl[i] = d13[i];
    for (int i = 0; i < 3; ++i) { dl[25 + i] = d13[7 + i]; dl[28 + i] = d13[10 + i]; }
    bool has_nan = false;
    for (int i = 0; i < 36; ++i) has_nan = has_nan || (vc[i] != vc[i]);
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            dl[7 + i + 3 * j] = has_nan ? 0.0 : vc[i + 6 * j] * dt * dt;                         // :171
            dl[16 + i + 3 * j] = has_nan ? 0.0 : vc[(3 + i) + 6 * (3 + j)] * dt * dt;            // :172
        }
    if (use_tf) {                                                                              // :202-215
        double c2[36], c1[36], t[7], c[36];
        for (int i = 0; i < 36; ++i) { c2[i] = 0.0; c1[i] = 0.0; }
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {                                                      // Transform.cpp:294-296
                c2[i + 6 * j] = pv[16 + i + 3 * j]; c2[(3 + i) + 6 * (3 + j)] = pv[7 + i + 3 * j];
                c1[i + 6 * j] = dl[16 + i + 3 * j]; c1[(3 + i) + 6 * (3 + j)] = dl[7 + i + 3 * j];
            }
        transform_compose(pv, c2, dl, c1, t, c);
        for (int i = 0; i < 7; ++i) po[i] = t[i];
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {                                                      // Transform.cpp:314-321
                po[16 + i + 3 * j] = c[i + 6 * j];
                po[7 + i + 3 * j] = c[(3 + i) + 6 * (3 + j)];
            }
    } else {                                                                                   // :216-223
        double rx, ry, rz;
        qrot(ldq(pv + 3), dl[0], dl[1], dl[2], rx, ry, rz);
        po[0] += rx; po[1] += ry; po[2] += rz;
        for (int i = 0; i < 9; ++i) { po[7 + i] += dl[7 + i]; po[16 + i] += dl[16 + i]; }
        stq(po + 3, qmul(ldq(pv + 3), ldq(dl + 3)));
    }
    for (int i = 0; i < 3; ++i) { po[25 + i] = uu[1 + i]; po[37 + i] = uu[4 + i]; }            // :226-229
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            po[28 + i + 3 * j] = vc[i + 6 * j];
            po[40 + i + 3 * j] = vc[(3 + i) + 6 * (3 + j)];
        }
    if (delta) for (int i = 0; i < 31; ++i) delta[(size_t)b * 31 + i] = dl[i];
}

// Singular values (decreasing) and left singular vectors of a symmetric positive semi-definite 3x3 matrix: what
// Eigen::JacobiSVD(Uk, ComputeThinU) returns for it (MeasurementModels.hpp:230-235), as a cyclic Jacobi eigenvalue
// iteration; column signs are immaterial downstream (u u^T, u^T M u).
__device__ inline void sym3_svd(const double *A, double *s, double *U)
{
    double a[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) a[i] = A[i];
    for (int sweep = 0; sweep < 12; ++sweep)
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[3 * p + q];
                if (apq == 0.0) continue;
                const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[3 * k + p], akq = a[3 * k + q];
                    a[3 * k + p] = c * akp - sn * akq;
                    a[3 * k + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[3 * p + k], aqk = a[3 * q + k];
                    a[3 * p + k] = c * apk - sn * aqk;
                    a[3 * q + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - sn * vkq;
                    V[3 * k + q] = sn * vkp + c * vkq;
                }
            }
    int idx[3] = {0, 1, 2};
    const double e[3] = {fabs(a[0]), fabs(a[4]), fabs(a[8])};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (e[idx[j]] < e[idx[j + 1]]) { const int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
    for (int c = 0; c < 3; ++c) {
        s[c] = e[idx[c]];
        for (int k = 0; k < 3; ++k) U[3 * k + c] = V[3 * k + idx[c]];
    }
}

// AdaptiveAttitudeCov::matrix (MeasurementModels.hpp:181-286) for a batch of independent objects.
// hist [B][m1][9] (row-major 3x3 each), r2count [B]; r1 = the (batch-wide) slot this call writes.
__global__ void adaptive_attitude_cov_kernel(int B, unsigned m1, unsigned m2, double gamma, double *hist, unsigned r1,
                                             unsigned *r2count, int n, const double *xk, const double *Pk, const double *z,
                                             const double *H, const double *R, int r_stride, double *Rout)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *x = xk + (size_t)b * n, *P = Pk + (size_t)b * n * n, *Hb = H + (size_t)b * 3 * n, *Rb = R + (size_t)b * r_stride;
    double *hb = hist + (size_t)b * m1 * 9;
    double res[3], Uk[9], fooR[9], s[3], U[9], mu[3], Qs[9];
    for (int i = 0; i < 3; ++i) {                                                    // z - H xk, :195
        double hx = 0;
        for (int j = 0; j < n; ++j) hx += Hb[i + 3 * j] * x[j];
        res[i] = z[(size_t)b * 3 + i] - hx;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) hb[9 * r1 + 3 * i + j] = res[i] * res[j];         // :195-197
    for (int i = 0; i < 9; ++i) { Uk[i] = 0; Qs[i] = 0; }
    for (unsigned h = 0; h < m1; ++h)                                                // :215-223
        for (int i = 0; i < 9; ++i) Uk[i] += hb[9 * h + i];
    for (int i = 0; i < 9; ++i) Uk[i] = Uk[i] / (double)m1;
    for (int i = 0; i < 3; ++i)                                                      // fooR = H Pk H^T + R, :225
        for (int j = 0; j < 3; ++j) {
            double sum = 0;
            for (int a = 0; a < n; ++a) {
                double hp = 0;
                for (int c = 0; c < n; ++c) hp += Hb[i + 3 * c] * P[c + n * a];
                sum += hp * Hb[j + 3 * a];
            }
            fooR[3 * i + j] = sum + Rb[i + 3 * j];
        }
    sym3_svd(Uk, s, U);                                                              // :230-235
    for (int c = 0; c < 3; ++c) {                                                    // :237-239
        double sum = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) sum += U[3 * i + c] * fooR[3 * i + j] * U[3 * j + c];
        mu[c] = sum;
    }
    double mx = s[0] - mu[0];
    for (int c = 1; c < 3; ++c) if (s[c] - mu[c] > mx) mx = s[c] - mu[c];
    bool use;
    if (mx > gamma) { r2count[b] = 0; use = true; }                                  // :245-258
    else { const unsigned r2 = r2count[b] + 1; r2count[b] = r2; use = r2 < m2; }     // :259-275
    if (use)
        for (int c = 0; c < 3; ++c) {
            const double w = s[c] - mu[c] > 0.0 ? s[c] - mu[c] : 0.0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Qs[3 * i + j] += w * U[3 * i + c] * U[3 * j + c];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rout[(size_t)b * 9 + i + 3 * j] = Rb[i + 3 * j] + Qs[3 * i + j];   // :284
}

} // namespace slk
