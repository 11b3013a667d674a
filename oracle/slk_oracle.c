/*
 * slk_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See slk_oracle.h for scope, conventions and the "parity unpinned" statement.
 *
 * Restates, in reference operation order, the fp64 algorithm of
 *   /root/reference/src/filters/Msckf.hpp   (predict :89-189, UKF update :196-277,
 *       sigma points :400-468, means :471-538, covariances :554-657,
 *       applyDelta :659-666, removeOutliers :688-754, chi2 gate :844-905)
 *   /root/reference/src/filters/Usckf.hpp   (ctor :90-103, predict :107-244,
 *       update :246-308, setMeasurement :322-389, cloning :391-433,
 *       sigma points :532-598, means :601-640, covariances :654-737)
 *   /root/reference/src/filters/State.hpp   (set/boxplus/boxminus/vectorize :137-669)
 *   /root/reference/src/filters/MtkWrap.hpp (operator+,-: :77-102, :160-228, :277-310)
 *
 * Third-party arithmetic not under /root/reference, restated from the published
 * algorithm (MTK, Rock package slam/mtk, manifest.xml:16, no version pinned;
 * Eigen3, src/CMakeLists.txt:26, no version pinned):
 *   MTK::SO3::exp/log, vect boxplus/boxminus, Eigen::LLT (unblocked, lower),
 *   Eigen PartialPivLU inverse, Eigen quaternion product / vector rotation /
 *   toRotationMatrix.
 */
#include "slk_oracle.h"

#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ====================================================================== */
/* layout                                                                 */
/* ====================================================================== */

#define MAXBLK 160
typedef struct { int so3; int toff; int soff; int len; } blk_t;

/* State.hpp:141-149 (State), :246-252 (SensorState), :384-396 (MultiState
 * tangent order), :567-588 (AugmentedState tangent order) */
static int layout_blocks(const slko_layout *lay, blk_t *b)
{
    int n = 0, s, c;
    int nstates = (lay->kind == SLKO_AUGMENTED) ? 3 : 1;
    for (s = 0; s < nstates; ++s) {
        int t0 = 12 * s, s0 = 13 * s;
        b[n++] = (blk_t){0, t0 + 0, s0 + 0, 3};
        b[n++] = (blk_t){1, t0 + 3, s0 + 3, 3};
        b[n++] = (blk_t){0, t0 + 6, s0 + 7, 3};
        b[n++] = (blk_t){0, t0 + 9, s0 + 10, 3};
    }
    if (lay->kind == SLKO_MULTI) {
        for (c = 0; c < lay->k; ++c) {
            b[n++] = (blk_t){0, 12 + 6 * c, 13 + 7 * c, 3};
            b[n++] = (blk_t){1, 12 + 6 * c + 3, 13 + 7 * c + 3, 3};
        }
    } else if (lay->kind == SLKO_AUGMENTED) {
        if (lay->nfk > 0)  b[n++] = (blk_t){0, 36, 39, lay->nfk};
        if (lay->nfkl > 0) b[n++] = (blk_t){0, 36 + lay->nfk, 39 + lay->nfk, lay->nfkl};
    }
    return n;
}

int slko_dof(const slko_layout *lay)
{
    if (lay->kind == SLKO_MULTI) return 12 + 6 * lay->k;          /* State.hpp:373-376 */
    if (lay->kind == SLKO_AUGMENTED) return 36 + lay->nfk + lay->nfkl; /* State.hpp:590-593 */
    return 12;
}

int slko_storage(const slko_layout *lay)
{
    if (lay->kind == SLKO_MULTI) return 13 + 7 * lay->k;
    if (lay->kind == SLKO_AUGMENTED) return 39 + lay->nfk + lay->nfkl;
    return 13;
}

/* ====================================================================== */
/* SO(3): MTK::SO3<double> semantics (third-party, restated)              */
/* ====================================================================== */

/* MTK mtkmath.hpp cos_sinc_sqrt: returns (cos(sqrt(x)), sin(sqrt(x))/sqrt(x)),
 * Taylor series below eps^(1/4). */
static void cos_sinc_sqrt(double x, double *c, double *s)
{
    static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
    const double taylor_0_bound = 2.220446049250313e-16;
    const double taylor_2_bound = sqrt(taylor_0_bound);
    const double taylor_n_bound = sqrt(taylor_2_bound);
    if (x >= taylor_n_bound) {
        double sx = sqrt(x);
        *c = cos(sx);
        *s = sin(sx) / sx;
        return;
    }
    double cosi = 1., sinc = 1.;
    double term = -1 / 2. * x;
    for (int i = 0; i < 3; ++i) {
        cosi += term;
        term *= inv[2 * i];
        sinc += term;
        term *= -inv[2 * i + 1] * x;
    }
    *c = cosi;
    *s = sinc;
}

/* MTK::SO3::exp(dvec, scale): w = cos(scale*|v|/2), vec = sinc(scale*|v|/2)*(scale/2)*v.
 * Used by State::set (State.hpp:179) and SO3::boxplus (State.hpp:189). */
void slko_so3_exp(const double v[3], double scale, double q[4])
{
    double half = scale / 2;
    double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double c, s;
    cos_sinc_sqrt(half * half * n2, &c, &s);
    double mult = s * half;
    q[0] = mult * v[0];
    q[1] = mult * v[1];
    q[2] = mult * v[2];
    q[3] = c;
}

/* MTK::SO3::log(q) = MTK::log(res, w, vec, scale=2, plus_minus_periodicity=true):
 * 2*atan(|vec|/w)/|vec| * vec, |vec| clamped to tolerance 1e-11.
 * Used by State::getVectorizedState (State.hpp:231) and SO3::boxminus. */
void slko_so3_log(const double q[4], double v[3])
{
    double nv = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < 1e-11) nv = 1e-11;
    double s = 2.0 / nv * atan(nv / q[3]);
    v[0] = s * q[0];
    v[1] = s * q[1];
    v[2] = s * q[2];
}

/* Eigen quaternion product a*b, coefficient order (x,y,z,w) */
void slko_quat_mul(const double a[4], const double b[4], double o[4])
{
    double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* Eigen QuaternionBase::_transformVector: v + w*(2 u x v) + u x (2 u x v) */
void slko_quat_rotate(const double q[4], const double v[3], double o[3])
{
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

static void quat_conj(const double q[4], double o[4])
{
    o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3];
}

/* SO3::boxplus: q <- q * exp(v, scale=1) */
static void so3_boxplus(const double q[4], const double v[3], double o[4])
{
    double d[4];
    slko_so3_exp(v, 1.0, d);
    slko_quat_mul(q, d, o);
}

/* SO3::boxminus: res = log(other.conjugate() * this) */
static void so3_boxminus(const double a[4], const double b[4], double v[3])
{
    double bc[4], r[4];
    quat_conj(b, bc);
    slko_quat_mul(bc, a, r);
    slko_so3_log(r, v);
}

/* ====================================================================== */
/* manifold state operations                                              */
/* ====================================================================== */

/* x [+] v : State::boxplus State.hpp:186-192, SensorState :286-290,
 * MultiState::boxplus(vector) :418-434; MtkWrap operator+ MtkWrap.hpp:77-90,181-195 */
void slko_boxplus(const slko_layout *lay, const double *x, const double *v, double *out)
{
    blk_t b[MAXBLK];
    int nb = layout_blocks(lay, b);
    for (int i = 0; i < nb; ++i) {
        if (b[i].so3) {
            so3_boxplus(x + b[i].soff, v + b[i].toff, out + b[i].soff);
        } else {
            for (int j = 0; j < b[i].len; ++j) out[b[i].soff + j] = x[b[i].soff + j] + 1.0 * v[b[i].toff + j];
        }
    }
}

/* a [-] b : State::boxminus State.hpp:194-200, MultiState::boxminus :460-481;
 * MtkWrap operator- MtkWrap.hpp:95-102, 218-228 */
void slko_boxminus(const slko_layout *lay, const double *a, const double *bb, double *out)
{
    blk_t b[MAXBLK];
    int nb = layout_blocks(lay, b);
    for (int i = 0; i < nb; ++i) {
        if (b[i].so3) {
            so3_boxminus(a + b[i].soff, bb + b[i].soff, out + b[i].toff);
        } else {
            for (int j = 0; j < b[i].len; ++j) out[b[i].toff + j] = a[b[i].soff + j] - bb[b[i].soff + j];
        }
    }
}

/* State::set(vstate, ANGLE_AXIS) State.hpp:166-184, MultiState::set :380-399,
 * AugmentedState::set :567-588 */
void slko_set_from_vector(const slko_layout *lay, const double *v, double *x)
{
    blk_t b[MAXBLK];
    int nb = layout_blocks(lay, b);
    for (int i = 0; i < nb; ++i) {
        if (b[i].so3) slko_so3_exp(v + b[i].toff, 1.0, x + b[i].soff);
        else for (int j = 0; j < b[i].len; ++j) x[b[i].soff + j] = v[b[i].toff + j];
    }
}

/* getVectorizedState(ANGLE_AXIS) State.hpp:215-239, :509-526, :648-668 */
void slko_vectorize(const slko_layout *lay, const double *x, double *v)
{
    blk_t b[MAXBLK];
    int nb = layout_blocks(lay, b);
    for (int i = 0; i < nb; ++i) {
        if (b[i].so3) slko_so3_log(x + b[i].soff, v + b[i].toff);
        else for (int j = 0; j < b[i].len; ++j) v[b[i].toff + j] = x[b[i].soff + j];
    }
}

static void identity_state(const slko_layout *lay, double *x)
{
    int nq = slko_storage(lay);
    blk_t b[MAXBLK];
    int nb = layout_blocks(lay, b);
    memset(x, 0, sizeof(double) * nq);
    for (int i = 0; i < nb; ++i) if (b[i].so3) x[b[i].soff + 3] = 1.0;
}

/* ====================================================================== */
/* dense helpers                                                          */
/* ====================================================================== */

#define AT(M, ld, i, j) (M)[(size_t)(j) * (ld) + (i)]

/* Eigen::LLT<Lower>, unblocked kernel (internal::llt_inplace<Scalar,Lower>::unblocked):
 * column k from the previously computed columns; stops at the first non-positive
 * pivot and leaves the remaining columns untouched.  The reference never reads
 * .info() (Usckf.hpp:537-538, :577-578; Msckf.hpp:412-413, :447-448).
 * Returns -1 on success, else the failing pivot index. matrixL() = lower triangle. */
int slko_cholesky_lower(int n, const double *A, double *L)
{
    int fail = -1;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i)
            AT(L, n, i, j) = (i >= j) ? AT(A, n, i, j) : 0.0;
    for (int k = 0; k < n; ++k) {
        double x = AT(L, n, k, k);
        for (int p = 0; p < k; ++p) x -= AT(L, n, k, p) * AT(L, n, k, p);
        if (!(x > 0.0)) { fail = k; break; }
        x = sqrt(x);
        AT(L, n, k, k) = x;
        for (int i = k + 1; i < n; ++i) {
            double s = AT(L, n, i, k);
            for (int p = 0; p < k; ++p) s -= AT(L, n, i, p) * AT(L, n, k, p);
            AT(L, n, i, k) = s / x;
        }
    }
    return fail;
}

/* Eigen MatrixBase::inverse() for n > 4: PartialPivLU, then solve against I
 * (Usckf.hpp:154, :286; Msckf.hpp:138, :257, :736). Returns 0 ok, 1 singular. */
int slko_inverse(int n, const double *A, double *Ainv)
{
    double *lu = (double *)malloc(sizeof(double) * n * n);
    int *perm = (int *)malloc(sizeof(int) * n);
    int singular = 0;
    memcpy(lu, A, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(AT(lu, n, k, k));
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(AT(lu, n, i, k));
            if (v > best) { best = v; piv = i; }
        }
        if (best == 0.0) { singular = 1; continue; }
        if (piv != k) {
            for (int j = 0; j < n; ++j) {
                double t = AT(lu, n, k, j); AT(lu, n, k, j) = AT(lu, n, piv, j); AT(lu, n, piv, j) = t;
            }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        for (int i = k + 1; i < n; ++i) AT(lu, n, i, k) /= AT(lu, n, k, k);
        for (int j = k + 1; j < n; ++j) {
            double ukj = AT(lu, n, k, j);
            for (int i = k + 1; i < n; ++i) AT(lu, n, i, j) -= AT(lu, n, i, k) * ukj;
        }
    }
    for (int c = 0; c < n; ++c) {
        /* solve L U x = P e_c */
        double *x = Ainv + (size_t)c * n;
        for (int i = 0; i < n; ++i) x[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i)
            for (int p = 0; p < i; ++p) x[i] -= AT(lu, n, i, p) * x[p];
        for (int i = n - 1; i >= 0; --i) {
            for (int p = i + 1; p < n; ++p) x[i] -= AT(lu, n, i, p) * x[p];
            x[i] /= AT(lu, n, i, i);
        }
    }
    free(lu);
    free(perm);
    return singular;
}

/* C(m x n) = A(m x k) * B(k x n), all column-major with given leading dims */
static void matmul(int m, int n, int k, const double *A, int lda, int ta, const double *B, int ldb, int tb,
                   double *C, int ldc)
{
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) {
            double s = 0;
            for (int p = 0; p < k; ++p) {
                double a = ta ? AT(A, lda, p, i) : AT(A, lda, i, p);
                double b = tb ? AT(B, ldb, j, p) : AT(B, ldb, p, j);
                s += a * b;
            }
            AT(C, ldc, i, j) = s;
        }
}

/* ====================================================================== */
/* sigma-point primitives shared by both filters                          */
/* ====================================================================== */

/* manifold mean: Usckf.hpp:601-627, Msckf.hpp:471-496 / :499-525.
 * X holds S states of storage size nq back to back. */
static int manifold_mean(const slko_layout *lay, const double *X, int S, double *ref, int *iters)
{
    int N = slko_dof(lay), nq = slko_storage(lay);
    const int max_it = 10000;
    double *mean_delta = (double *)malloc(sizeof(double) * N);
    double *d = (double *)malloc(sizeof(double) * N);
    double *tmp = (double *)malloc(sizeof(double) * nq);
    double norm;
    int i = 0;
    memcpy(ref, X, sizeof(double) * nq);
    do {
        for (int t = 0; t < N; ++t) mean_delta[t] = 0;
        for (int p = 0; p < S; ++p) {
            slko_boxminus(lay, X + (size_t)p * nq, ref, d);
            for (int t = 0; t < N; ++t) mean_delta[t] += d[t];
        }
        for (int t = 0; t < N; ++t) mean_delta[t] /= (double)S;
        slko_boxplus(lay, ref, mean_delta, tmp);
        memcpy(ref, tmp, sizeof(double) * nq);
        norm = 0;
        for (int t = 0; t < N; ++t) norm += mean_delta[t] * mean_delta[t];
        norm = sqrt(norm);
    } while (norm > 1e-6 && ++i < max_it);
    if (iters) *iters = i + 1;
    free(mean_delta); free(d); free(tmp);
    return (i >= max_it) ? SLKO_MEAN_NOT_CONVERGED : SLKO_OK;
}

/* covSigmaPoints: c = sum d d^T ; return 0.5 c  (Usckf.hpp:654-670, Msckf.hpp:574-589) */
static void cov_manifold(const slko_layout *lay, const double *mean, const double *X, int S, double *C)
{
    int N = slko_dof(lay), nq = slko_storage(lay);
    double *d = (double *)malloc(sizeof(double) * N);
    memset(C, 0, sizeof(double) * N * N);
    for (int p = 0; p < S; ++p) {
        slko_boxminus(lay, X + (size_t)p * nq, mean, d);
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) AT(C, N, i, j) += d[i] * d[j];
    }
    for (int i = 0; i < N * N; ++i) C[i] = 0.5 * C[i];
    free(d);
}

/* vector mean: accumulate / size (Usckf.hpp:630-640, Msckf.hpp:528-538) */
static void mean_vectors(const double *Z, int m, int S, double *zbar)
{
    for (int r = 0; r < m; ++r) zbar[r] = 0;
    for (int p = 0; p < S; ++p)
        for (int r = 0; r < m; ++r) zbar[r] += Z[(size_t)p * m + r];
    for (int r = 0; r < m; ++r) zbar[r] /= (double)S;
}

/* covSigmaPoints for measurement vectors (Usckf.hpp:672-689, Msckf.hpp:593-610) */
static void cov_vectors(const double *zbar, const double *Z, int m, int S, double *C)
{
    double *d = (double *)malloc(sizeof(double) * m);
    memset(C, 0, sizeof(double) * m * m);
    for (int p = 0; p < S; ++p) {
        for (int r = 0; r < m; ++r) d[r] = Z[(size_t)p * m + r] - zbar[r];
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i) AT(C, m, i, j) += d[i] * d[j];
    }
    for (int i = 0; i < m * m; ++i) C[i] = 0.5 * C[i];
    free(d);
}

/* Msckf.hpp:844-905 / Usckf.hpp:794-855: chi-square 95% gate, dof 1..9 only */
int slko_accept_mahalanobis(double d2, int dof)
{
    static const double thr[10] = {0, 3.84, 5.99, 7.81, 9.49, 11.07, 12.59, 14.07, 15.51, 16.92};
    if (dof < 1 || dof > 9) return 0;
    return d2 < thr[dof] ? 1 : 0;
}

/* ====================================================================== */
/* models                                                                 */
/* ====================================================================== */

/* test/UsckfUnitTest.cpp:34-49 */
void slko_pm_const_velocity(const double *x, double *y, void *ctx)
{
    const slko_const_velocity *p = (const slko_const_velocity *)ctx;
    double sa[3] = { p->angular_velocity[0] * p->dt, p->angular_velocity[1] * p->dt, p->angular_velocity[2] * p->dt };
    double rot[4];
    slko_so3_exp(sa, 1.0, rot);
    slko_quat_mul(x + 3, rot, y + 3);                              /* s2.orient = state.orient * rot */
    for (int i = 0; i < 3; ++i) y[10 + i] = p->angular_velocity[i]; /* s2.angvelo */
    for (int i = 0; i < 3; ++i) y[7 + i] = p->velocity[i];          /* s2.velo */
    for (int i = 0; i < 3; ++i) y[i] = x[i] + x[7 + i] * p->dt;     /* s2.pos = state.pos + state.velo*dt */
}

/* test/MsckfUnitTest.cpp:33-47 */
void slko_pm_delta_pose(const double *x, double *y, void *ctx)
{
    const slko_delta_pose *p = (const slko_delta_pose *)ctx;
    double r[3];
    slko_quat_mul(x + 3, p->dquat, y + 3);                          /* s2.orient = state.orient * delta_orientation */
    for (int i = 0; i < 3; ++i) y[10 + i] = p->angular_velocity[i];
    slko_quat_rotate(y + 3, p->dpos, r);                            /* s2.orient * delta_position */
    for (int i = 0; i < 3; ++i) y[i] = x[i] + r[i];
    for (int i = 0; i < 3; ++i) y[7 + i] = p->velocity[i];
}

/* src/core/DeadReckon.hpp:246-286 (updateAttitude): third-order quaternion integration of the angular
 * velocity, constant angular acceleration between the previous (w1) and the current (w0) sample.
 * The 4x4 expression of the reference acts on the identity quaternion (w,x,y,z) = (1,0,0,0): only its
 * first column is ever used, written out here.  Result (x,y,z,w), normalised like Eigen::normalize. */
void slko_update_attitude(double dt, const double w0[3], const double w1[3], double q[4])
{
    const double n2 = w0[0] * w0[0] + w0[1] * w0[1] + w0[2] * w0[2];
    const double dot = w0[0] * w1[0] + w0[1] * w1[1] + w0[2] * w1[2];
    const double cr[3] = { w0[1] * w1[2] - w0[2] * w1[1], w0[2] * w1[0] - w0[0] * w1[2], w0[0] * w1[1] - w0[1] * w1[0] };
    const double dt2 = dt * dt, dt3 = dt2 * dt;
    /* (omega4 * oldomega4) e0 = (-w0.w1, -(w0 x w1)) */
    double qw = 1.0 - (1.0 / 6.0) * n2 * dt2 - (1.0 / 24.0) * (-dot) * dt2;
    double qv[3];
    for (int i = 0; i < 3; ++i)
        qv[i] = 0.75 * w0[i] * dt - 0.25 * w1[i] * dt - (1.0 / 24.0) * (-cr[i]) * dt2 - (1.0 / 48.0) * n2 * w0[i] * dt3;
    const double nrm = sqrt(qw * qw + qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2]);
    q[0] = qv[0] / nrm; q[1] = qv[1] / nrm; q[2] = qv[2] / nrm; q[3] = qw / nrm;
}

/* src/core/DeadReckon.hpp:129-239, the delta pose of updatePose(delta_t, cartesianVelocities, ...):
 * u = dt, v0[3], w0[3] (current linear / angular velocity), v1[3], w1[3] (previous sample).
 * delta = dpos[3] dquat[4] velocity[3] angular_velocity[3] = the input of the delta-pose process model. */
void slko_dead_reckon_delta(const double u[13], double delta[13])
{
    const double dt = u[0];
    for (int i = 0; i < 3; ++i) delta[i] = (dt / 2.0) * (u[1 + i] + u[7 + i]);      /* :148 */
    slko_update_attitude(dt, u + 4, u + 10, delta + 3);                              /* :161 */
    for (int i = 0; i < 3; ++i) delta[7 + i] = u[1 + i];                             /* :150 */
    for (int i = 0; i < 3; ++i) delta[10 + i] = u[4 + i];                            /* :151 */
}

/* dead reckoning feeding the delta-pose process model (test/MsckfUnitTest.cpp:33-47); ctx = double[13] u */
void slko_pm_dead_reckon(const double *x, double *y, void *ctx)
{
    double d[13];
    slko_delta_pose p;
    slko_dead_reckon_delta((const double *)ctx, d);
    memcpy(p.dpos, d, 3 * sizeof(double));
    memcpy(p.dquat, d + 3, 4 * sizeof(double));
    memcpy(p.velocity, d + 7, 3 * sizeof(double));
    memcpy(p.angular_velocity, d + 10, 3 * sizeof(double));
    slko_pm_delta_pose(x, y, &p);
}

/* Eigen QuaternionBase::toRotationMatrix (used through Eigen::Affine3d(orient),
 * UsckfUnitTest.cpp:71). R is 3x3 column-major. */
static void quat_to_matrix(const double q[4], double R[9])
{
    double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    AT(R, 3, 0, 0) = 1 - (tyy + tzz); AT(R, 3, 0, 1) = txy - twz; AT(R, 3, 0, 2) = txz + twy;
    AT(R, 3, 1, 0) = txy + twz; AT(R, 3, 1, 1) = 1 - (txx + tzz); AT(R, 3, 1, 2) = tyz - twx;
    AT(R, 3, 2, 0) = txz - twy; AT(R, 3, 2, 1) = tyz + twx; AT(R, 3, 2, 2) = 1 - (txx + tyy);
}

/* test/UsckfUnitTest.cpp:62-86: featuresk transformed by (statek - statek_i) */
void slko_mm_vo_relative(const slko_layout *lay, const double *X, int m, double *z, void *ctx)
{
    (void)ctx;
    slko_layout single = {SLKO_SINGLE, 0, 0, 0};
    double d[12], ds[13], R[9];
    const double *statek = X, *statek_i = X + 26, *fk = X + 39;
    (void)lay;
    slko_boxminus(&single, statek, statek_i, d);   /* delta_state = statek - statek_i (vector) */
    slko_set_from_vector(&single, d, ds);          /* ... assigned to a WSingleState: set() */
    quat_to_matrix(ds + 3, R);
    for (int i = 0; i < m; ++i) z[i] = fk[i];      /* z_hat = wastate.featuresk */
    for (int i = 0; i + 2 < m; i += 3) {
        double c[3] = { fk[i], fk[i + 1], fk[i + 2] };
        for (int r = 0; r < 3; ++r)
            z[i + r] = (AT(R, 3, r, 0) * c[0] + AT(R, 3, r, 1) * c[1] + AT(R, 3, r, 2) * c[2]) + ds[r];
    }
}

static void pose_of(const slko_layout *lay, const double *X, int c, const double **p, const double **q)
{
    if (lay->kind == SLKO_MULTI) {
        if (c == 0) { *p = X; *q = X + 3; }
        else { *p = X + 13 + 7 * (c - 1); *q = *p + 3; }
    } else if (lay->kind == SLKO_AUGMENTED) {
        *p = X + 13 * c; *q = *p + 3;
    } else { *p = X; *q = X + 3; }
}

/* Registered Msckf measurement model of this build (the reference ships no
 * Msckf measurement model: MsckfUnitTest.cpp never calls update()).  Feature j is
 * a landmark l_j observed from pose c_j as a normalised image point:
 *   z_j = ( x/z, y/z ),  (x,y,z) = R(q_c)^T (l_j - p_c). */
void slko_mm_feature_proj(const slko_layout *lay, const double *X, int m, double *z, void *ctx)
{
    const double *f = (const double *)ctx;
    for (int j = 0; j < m / 2; ++j) {
        const double *p, *q;
        double qc[4], v[3], loc[3];
        pose_of(lay, X, (int)f[4 * j + 3], &p, &q);
        for (int i = 0; i < 3; ++i) v[i] = f[4 * j + i] - p[i];
        quat_conj(q, qc);
        slko_quat_rotate(qc, v, loc);
        z[2 * j] = loc[0] / loc[2];
        z[2 * j + 1] = loc[1] / loc[2];
    }
}

void slko_mm_pose_position(const slko_layout *lay, const double *X, int m, double *z, void *ctx)
{
    const double *p, *q;
    pose_of(lay, X, (int)((const double *)ctx)[0], &p, &q);
    for (int i = 0; i < m && i < 3; ++i) z[i] = p[i];
}

/* ====================================================================== */
/* single-state predict core shared by Usckf::predict and Msckf::predict  */
/* ====================================================================== */

/* generateSigmaPoints(single) Usckf.hpp:565-598 / Msckf.hpp:435-468 */
static int sigma_points_vec(const slko_layout *lay, const double *mu, const double *delta, const double *P,
                            double *X, double *Lout)
{
    int N = slko_dof(lay), nq = slko_storage(lay);
    double *L = Lout ? Lout : (double *)malloc(sizeof(double) * N * N);
    double *v = (double *)malloc(sizeof(double) * N);
    int fail = slko_cholesky_lower(N, P, L);
    for (int t = 0; t < N; ++t) v[t] = delta ? delta[t] : 0.0;
    slko_boxplus(lay, mu, v, X);                                   /* X[0] = mu + delta */
    for (int j = 0; j < N; ++j) {
        for (int t = 0; t < N; ++t) v[t] = (delta ? delta[t] : 0.0) + AT(L, N, t, j);
        slko_boxplus(lay, mu, v, X + (size_t)(2 * j + 1) * nq);    /* mu + (delta + L.col(j)) */
        for (int t = 0; t < N; ++t) v[t] = (delta ? delta[t] : 0.0) - AT(L, N, t, j);
        slko_boxplus(lay, mu, v, X + (size_t)(2 * j + 2) * nq);    /* mu + (delta - L.col(j)) */
    }
    free(v);
    if (!Lout) free(L);
    return fail >= 0 ? SLKO_LLT_FAIL : SLKO_OK;
}

/* Usckf.hpp:117-178 == Msckf.hpp:102-162: sigma points of the 12-DOF current
 * state, process model map, manifold mean, Pxy, Fk, new Pk_i = cov + Q. */
static int predict_single(double *statek_i /*13, in/out*/, double *Pk_i /*12x12 in/out*/,
                          slko_process_fn fn, void *ctx, const double *Q, double *Fk /*12x12 out*/, int *iters)
{
    const slko_layout single = {SLKO_SINGLE, 0, 0, 0};
    enum { n = 12, nq = 13, S = 25 };
    double X[S * nq], XCopy[S * nq], Y[nq], mean_new[nq], Pxy[n * n], Pinv[n * n], C[n * n];
    double dx[n], dy[n];
    int status = sigma_points_vec(&single, statek_i, NULL, Pk_i, X, NULL);
    memcpy(XCopy, X, sizeof(X));
    for (int p = 0; p < S; ++p) {                       /* std::transform(X, X, f) */
        fn(XCopy + p * nq, Y, ctx);
        memcpy(X + p * nq, Y, sizeof(Y));
    }
    status |= manifold_mean(&single, X, S, mean_new, iters);
    /* crossCovSigmaPoints(statek_i_old, mean_new, XCopy, X): Usckf.hpp:691-712 */
    memset(Pxy, 0, sizeof(Pxy));
    for (int p = 0; p < S; ++p) {
        slko_boxminus(&single, XCopy + p * nq, statek_i, dx);
        slko_boxminus(&single, X + p * nq, mean_new, dy);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) AT(Pxy, n, i, j) += dx[i] * dy[j];
    }
    for (int i = 0; i < n * n; ++i) Pxy[i] = 0.5 * Pxy[i];
    if (slko_inverse(n, Pk_i, Pinv)) status |= SLKO_SINGULAR;
    matmul(n, n, n, Pxy, n, 1, Pinv, n, 0, Fk, n);      /* Fk = Pxy^T * Pk_i^-1 */
    cov_manifold(&single, mean_new, X, S, C);
    for (int i = 0; i < n * n; ++i) Pk_i[i] = C[i] + Q[i]; /* Pk_i = cov + Qk */
    memcpy(statek_i, mean_new, sizeof(mean_new));
    return status;
}

/* ====================================================================== */
/* Msckf                                                                  */
/* ====================================================================== */

slko_msckf *slko_msckf_new(int k, const double *mean, const double *P)
{
    slko_msckf *f = (slko_msckf *)calloc(1, sizeof(*f));
    f->lay = (slko_layout){SLKO_MULTI, k, 0, 0};
    int N = slko_dof(&f->lay), nq = slko_storage(&f->lay);
    f->mean = (double *)malloc(sizeof(double) * nq);
    f->P = (double *)malloc(sizeof(double) * N * N);
    memcpy(f->mean, mean, sizeof(double) * nq);   /* Msckf.hpp:80-85 */
    memcpy(f->P, P, sizeof(double) * N * N);
    return f;
}

void slko_msckf_free(slko_msckf *f)
{
    if (!f) return;
    free(f->mean); free(f->P); free(f);
}

/* Msckf::predict, Msckf.hpp:89-189.  Only the 12x12 leading block and the
 * 12-DOF mean change; the state<->clone cross-covariances are left stale
 * because the propagation is commented out in the reference (:171-182). */
int slko_msckf_predict(slko_msckf *f, slko_process_fn fn, void *ctx, const double *Q)
{
    int N = slko_dof(&f->lay);
    double Pk_i[144];
    for (int j = 0; j < 12; ++j)
        for (int i = 0; i < 12; ++i) AT(Pk_i, 12, i, j) = AT(f->P, N, i, j);
    int status = predict_single(f->mean, Pk_i, fn, ctx, Q, f->Fk, &f->mean_iters);
    for (int j = 0; j < 12; ++j)
        for (int i = 0; i < 12; ++i) AT(f->P, N, i, j) = AT(Pk_i, 12, i, j);
    return status;
}

/* removeRow / removeColumn semantics of Msckf.hpp:688-721 on an index list:
 * remove position `pos`; if pos is past the end the LAST entry is dropped
 * (conservativeResize without a shift). */
static void idx_remove(int *idx, int *count, int pos)
{
    int numRows = *count - 1;
    if (pos < numRows)
        for (int i = pos; i < numRows; ++i) idx[i] = idx[i + 1];
    *count = numRows;
}

/* Msckf::update (UKF), Msckf.hpp:196-277, with removeOutliers :723-754 and
 * applyDelta :659-666.  gate != 0 uses accept_mahalanobis_distance (:199),
 * gate == 0 accepts every block (a caller-supplied mt). */
int slko_msckf_update(slko_msckf *f, const double *z, int m, slko_measure_fn h, void *ctx,
                      const double *R, int gate, unsigned *n_outliers)
{
    const slko_layout *lay = &f->lay;
    int N = slko_dof(lay), nq = slko_storage(lay), S = 2 * N + 1;
    int status = SLKO_OK;
    double *X = (double *)malloc(sizeof(double) * (size_t)S * nq);
    double *Z = (double *)malloc(sizeof(double) * (size_t)S * m);
    double *zbar = (double *)malloc(sizeof(double) * m);
    double *innov = (double *)malloc(sizeof(double) * m);
    double *Sm = (double *)malloc(sizeof(double) * m * m);
    double *covXZ = (double *)malloc(sizeof(double) * N * m);
    double *d = (double *)malloc(sizeof(double) * N);
    int *idx = (int *)malloc(sizeof(int) * (m + 2));
    unsigned outliers = 0;

    status |= sigma_points_vec(lay, f->mean, NULL, f->P, X, NULL);      /* :228-229 */
    for (int p = 0; p < S; ++p) h(lay, X + (size_t)p * nq, m, Z + (size_t)p * m, ctx); /* :231-232 */
    mean_vectors(Z, m, S, zbar);                                         /* :234 */
    for (int r = 0; r < m; ++r) innov[r] = z[r] - zbar[r];               /* :236 */
    cov_vectors(zbar, Z, m, S, Sm);
    for (int i = 0; i < m * m; ++i) Sm[i] += R[i];                       /* :238 */
    memset(covXZ, 0, sizeof(double) * N * m);                            /* :239 -> :635-657 */
    for (int p = 0; p < S; ++p) {
        slko_boxminus(lay, X + (size_t)p * nq, f->mean, d);
        for (int j = 0; j < m; ++j) {
            double dz = Z[(size_t)p * m + j] - zbar[j];
            for (int i = 0; i < N; ++i) AT(covXZ, N, i, j) += d[i] * dz;
        }
    }
    for (int i = 0; i < N * m; ++i) covXZ[i] = 0.5 * covXZ[i];

    /* removeOutliers(innovation, covXZ, S, mt, dof = 2): :723-754.  The erased
     * positions are dof*i and then dof*i+1 AFTER the first erase shifted the
     * rows, i.e. original rows 2i and 2i+2 (reference quirk, reproduced). */
    int cnt = m;
    for (int r = 0; r < m; ++r) idx[r] = r;
    {
        const int dof = 2;
        int i = 0;
        while (i < cnt / dof) {
            int a = idx[dof * i], b = idx[dof * i + 1];
            double s00 = AT(Sm, m, a, a), s01 = AT(Sm, m, a, b), s10 = AT(Sm, m, b, a), s11 = AT(Sm, m, b, b);
            double blk[4] = {s00, s10, s01, s11}, inv[4];
            slko_inverse(2, blk, inv);
            double r0 = innov[a], r1 = innov[b];
            double d2 = r0 * (inv[0] * r0 + inv[2] * r1) + r1 * (inv[1] * r0 + inv[3] * r1);
            int ok = gate ? slko_accept_mahalanobis(d2, dof) : 1;
            if (!ok) {
                idx_remove(idx, &cnt, dof * i);
                idx_remove(idx, &cnt, dof * i + 1);
                outliers++;
            } else {
                i++;
            }
        }
    }

    if (cnt > 0) {                                                        /* :250 */
        int mm = cnt;
        double *Sr = (double *)malloc(sizeof(double) * mm * mm);
        double *Sinv = (double *)malloc(sizeof(double) * mm * mm);
        double *Cr = (double *)malloc(sizeof(double) * N * mm);
        double *K = (double *)malloc(sizeof(double) * N * mm);
        double *KS = (double *)malloc(sizeof(double) * N * mm);
        double *KSKt = (double *)malloc(sizeof(double) * N * N);
        double *delta = (double *)malloc(sizeof(double) * N);
        double *mean_new = (double *)malloc(sizeof(double) * nq);
        for (int j = 0; j < mm; ++j) {
            for (int i = 0; i < mm; ++i) AT(Sr, mm, i, j) = AT(Sm, m, idx[i], idx[j]);
            for (int i = 0; i < N; ++i) AT(Cr, N, i, j) = AT(covXZ, N, i, idx[j]);
        }
        if (slko_inverse(mm, Sr, Sinv)) status |= SLKO_SINGULAR;
        matmul(N, mm, mm, Cr, N, 0, Sinv, mm, 0, K, N);                  /* K = covXZ * S^-1  :257 */
        matmul(N, mm, mm, K, N, 0, Sr, mm, 0, KS, N);
        matmul(N, N, mm, KS, N, 0, K, N, 1, KSKt, N);
        for (int i = 0; i < N * N; ++i) f->P[i] -= KSKt[i];              /* Pk -= K S K^T   :262 */
        for (int i = 0; i < N; ++i) {                                    /* K * innovation  :263 */
            double s = 0;
            for (int j = 0; j < mm; ++j) s += AT(K, N, i, j) * innov[idx[j]];
            delta[i] = s;
        }
        /* applyDelta(delta): :659-666 */
        status |= sigma_points_vec(lay, f->mean, delta, f->P, X, NULL);
        status |= manifold_mean(lay, X, S, mean_new, &f->mean_iters);
        memcpy(f->mean, mean_new, sizeof(double) * nq);
        cov_manifold(lay, f->mean, X, S, f->P);
        free(Sr); free(Sinv); free(Cr); free(K); free(KS); free(KSKt); free(delta); free(mean_new);
    }
    if (n_outliers) *n_outliers = outliers;
    free(X); free(Z); free(zbar); free(innov); free(Sm); free(covXZ); free(d); free(idx);
    return status;
}

/* Eigen::HouseholderQR (unblocked sweep; the blocked variant Eigen picks for >= 48 columns applies the same
 * reflectors in compact WY form).  A (rows x cols, column-major, ld rows) is overwritten with R in its upper
 * triangle and the essential parts of the reflectors below it; tau[min(rows, cols)].
 * makeHouseholder: beta = -sign(c0) * ||x||, tau = (beta - c0) / beta, v = (1, tail / (c0 - beta)); an exactly
 * zero tail gives tau = 0, beta = c0. */
static void householder_qr(int rows, int cols, double *A, double *tau)
{
    int p = rows < cols ? rows : cols;
    for (int k = 0; k < p; ++k) {
        double c0 = AT(A, rows, k, k), tail = 0.0;
        for (int i = k + 1; i < rows; ++i) tail += AT(A, rows, i, k) * AT(A, rows, i, k);
        double beta, t;
        if (tail <= DBL_MIN) {
            t = 0.0; beta = c0;
            for (int i = k + 1; i < rows; ++i) AT(A, rows, i, k) = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            for (int i = k + 1; i < rows; ++i) AT(A, rows, i, k) /= (c0 - beta);
            t = (beta - c0) / beta;
        }
        AT(A, rows, k, k) = beta;
        tau[k] = t;
        for (int j = k + 1; j < cols; ++j) {            /* A[k:, j] -= tau v (v^T A[k:, j]) */
            double w = AT(A, rows, k, j);
            for (int i = k + 1; i < rows; ++i) w += AT(A, rows, i, k) * AT(A, rows, i, j);
            w *= t;
            AT(A, rows, k, j) -= w;
            for (int i = k + 1; i < rows; ++i) AT(A, rows, i, j) -= AT(A, rows, i, k) * w;
        }
    }
}

/* thinQ = householderQ() * Identity(rows, cols): Msckf.hpp:802-803 */
static void householder_thin_q(int rows, int cols, const double *A, const double *tau, double *Q)
{
    int p = rows < cols ? rows : cols;
    for (int j = 0; j < cols; ++j)
        for (int i = 0; i < rows; ++i) AT(Q, rows, i, j) = (i == j) ? 1.0 : 0.0;
    for (int k = p - 1; k >= 0; --k)
        for (int j = 0; j < cols; ++j) {
            double w = AT(Q, rows, k, j);
            for (int i = k + 1; i < rows; ++i) w += AT(A, rows, i, k) * AT(Q, rows, i, j);
            w *= tau[k];
            AT(Q, rows, k, j) -= w;
            for (int i = k + 1; i < rows; ++i) AT(Q, rows, i, j) -= AT(A, rows, i, k) * w;
        }
}

/* Msckf EKF update, Msckf.hpp:284-349: z, zmean = h(mu) and the Jacobian H (m x N, column-major) come from the
 * caller's functor (:310); removeOutliers :756-789 (information matrix inverted ONCE, its 2x2 blocks indexed with the
 * running i; rows erased with the same shifted second erase as the UKF overload), reduceDimension :791-816
 * (needs m' >= N rows), gain and covariance :332-337, state correction by boxplus, guaranteeSPD's result is
 * discarded (:340).  Returns status bits; SLKO_EKF_ROWS when fewer than N rows survive. */
int slko_msckf_update_ekf(slko_msckf *f, const double *z, const double *zmean, const double *H, int m,
                          const double *R, int gate, unsigned *n_outliers)
{
    const slko_layout *lay = &f->lay;
    int N = slko_dof(lay), nq = slko_storage(lay);
    int status = SLKO_OK;
    unsigned outliers = 0;
    double *innov = (double *)malloc(sizeof(double) * m);
    double *PHt = (double *)malloc(sizeof(double) * (size_t)N * m);
    double *S0 = (double *)malloc(sizeof(double) * (size_t)m * m);
    double *info = (double *)malloc(sizeof(double) * (size_t)m * m);
    int *idx = (int *)malloc(sizeof(int) * (m + 2));
    for (int r = 0; r < m; ++r) innov[r] = z[r] - zmean[r];                       /* :312 */
    matmul(N, m, N, f->P, N, 0, H, m, 1, PHt, N);                                 /* P H^T */
    matmul(m, m, N, H, m, 0, PHt, N, 0, S0, m);
    for (int i = 0; i < m * m; ++i) S0[i] += R[i];
    if (slko_inverse(m, S0, info)) status |= SLKO_SINGULAR;                       /* :765-766 */
    int cnt = m;
    for (int r = 0; r < m; ++r) idx[r] = r;
    {
        const int dof = 2;
        int i = 0;
        while (i < cnt / dof) {                                                   /* :771-787 */
            double r0 = innov[idx[dof * i]], r1 = innov[idx[dof * i + 1]];
            int a = dof * i, b = dof * i + 1;                                     /* block of the UNREDUCED information matrix */
            double d2 = r0 * (AT(info, m, a, a) * r0 + AT(info, m, a, b) * r1)
                      + r1 * (AT(info, m, b, a) * r0 + AT(info, m, b, b) * r1);
            int ok = gate ? slko_accept_mahalanobis(d2, dof) : 1;
            if (!ok) {
                idx_remove(idx, &cnt, dof * i);
                idx_remove(idx, &cnt, dof * i + 1);
                outliers++;
            } else {
                i++;
            }
        }
    }
    if (cnt > 0 && cnt < N) {
        status |= SLKO_EKF_ROWS;              /* reduceDimension would read R.block(0,0,N,N) out of range (:806) */
    } else if (cnt > 0) {                                                         /* :321 */
        int mm = cnt;
        double *Hq = (double *)malloc(sizeof(double) * (size_t)mm * N);
        double *tq = (double *)malloc(sizeof(double) * N);
        double *Q1 = (double *)malloc(sizeof(double) * (size_t)mm * N);
        double *Rr = (double *)malloc(sizeof(double) * (size_t)mm * mm);
        double *T1 = (double *)malloc(sizeof(double) * (size_t)mm * N);
        double *Hr = (double *)calloc((size_t)N * N, sizeof(double));
        double *Rn = (double *)malloc(sizeof(double) * (size_t)N * N);
        double *rn = (double *)malloc(sizeof(double) * N);
        double *T2 = (double *)malloc(sizeof(double) * (size_t)N * N);
        double *S = (double *)malloc(sizeof(double) * (size_t)N * N);
        double *Sinv = (double *)malloc(sizeof(double) * (size_t)N * N);
        double *K = (double *)malloc(sizeof(double) * (size_t)N * N);
        double *KS = (double *)malloc(sizeof(double) * (size_t)N * N);
        double *KSKt = (double *)malloc(sizeof(double) * (size_t)N * N);
        double *delta = (double *)malloc(sizeof(double) * N);
        double *mean_new = (double *)malloc(sizeof(double) * nq);
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < mm; ++i) AT(Hq, mm, i, j) = AT(H, m, idx[i], j);
        for (int j = 0; j < mm; ++j)
            for (int i = 0; i < mm; ++i) AT(Rr, mm, i, j) = AT(R, m, idx[i], idx[j]);
        householder_qr(mm, N, Hq, tq);                                            /* :797 */
        householder_thin_q(mm, N, Hq, tq, Q1);                                    /* :802-803 */
        for (int j = 0; j < N; ++j)
            for (int i = 0; i <= j; ++i) AT(Hr, N, i, j) = AT(Hq, mm, i, j);      /* :806, upper triangle of R */
        for (int j = 0; j < N; ++j) {                                             /* :809 */
            double sacc = 0;
            for (int i = 0; i < mm; ++i) sacc += AT(Q1, mm, i, j) * innov[idx[i]];
            rn[j] = sacc;
        }
        matmul(mm, N, mm, Rr, mm, 0, Q1, mm, 0, T1, mm);                          /* :812 */
        matmul(N, N, mm, Q1, mm, 1, T1, mm, 0, Rn, N);
        matmul(N, N, N, f->P, N, 0, Hr, N, 1, T2, N);                             /* P H^T */
        matmul(N, N, N, Hr, N, 0, T2, N, 0, S, N);
        for (int i = 0; i < N * N; ++i) S[i] += Rn[i];                            /* :324 */
        if (slko_inverse(N, S, Sinv)) status |= SLKO_SINGULAR;
        matmul(N, N, N, T2, N, 0, Sinv, N, 0, K, N);                              /* :325 */
        matmul(N, N, N, K, N, 0, S, N, 0, KS, N);
        matmul(N, N, N, KS, N, 0, K, N, 1, KSKt, N);
        for (int i = 0; i < N * N; ++i) f->P[i] -= KSKt[i];                       /* :330 */
        for (int i = 0; i < N; ++i) {
            double sacc = 0;
            for (int j = 0; j < N; ++j) sacc += AT(K, N, i, j) * rn[j];
            delta[i] = sacc;
        }
        slko_boxplus(lay, f->mean, delta, mean_new);                              /* :331 */
        memcpy(f->mean, mean_new, sizeof(double) * nq);
        free(Hq); free(tq); free(Q1); free(Rr); free(T1); free(Hr); free(Rn); free(rn); free(T2); free(S); free(Sinv);
        free(K); free(KS); free(KSKt); free(delta); free(mean_new);
    }
    if (n_outliers) *n_outliers = outliers;
    free(innov); free(PHt); free(S0); free(info); free(idx);
    return status;
}

/* Msckf::checkSigmaPoints, Msckf.hpp:819-839 (used as a KAT definition) */
int slko_msckf_check_sigma_points(const slko_msckf *f, double *max_cov_err, double *mean_err)
{
    const slko_layout *lay = &f->lay;
    int N = slko_dof(lay), nq = slko_storage(lay), S = 2 * N + 1;
    double *X = (double *)malloc(sizeof(double) * (size_t)S * nq);
    double *mu = (double *)malloc(sizeof(double) * nq);
    double *C = (double *)malloc(sizeof(double) * N * N);
    double *d = (double *)malloc(sizeof(double) * N);
    int status = sigma_points_vec(lay, f->mean, NULL, f->P, X, NULL);
    status |= manifold_mean(lay, X, S, mu, NULL);
    cov_manifold(lay, mu, X, S, C);
    double e = 0;
    for (int i = 0; i < N * N; ++i) { double a = fabs(C[i] - f->P[i]); if (a > e) e = a; }
    slko_boxminus(lay, f->mean, mu, d);
    double n2 = 0;
    for (int i = 0; i < N; ++i) n2 += d[i] * d[i];
    *max_cov_err = e;
    *mean_err = sqrt(n2);
    free(X); free(mu); free(C); free(d);
    return status;
}

/* ====================================================================== */
/* Usckf                                                                  */
/* ====================================================================== */

static void usckf_resize(slko_usckf *f, int nfk, int nfkl)
{
    f->lay.nfk = nfk; f->lay.nfkl = nfkl;
}

slko_usckf *slko_usckf_new(int nfk, int nfkl, const double *mean, const double *P)
{
    slko_usckf *f = (slko_usckf *)calloc(1, sizeof(*f));
    f->lay = (slko_layout){SLKO_AUGMENTED, 0, nfk, nfkl};
    int N = slko_dof(&f->lay), nq = slko_storage(&f->lay);
    f->mean = (double *)malloc(sizeof(double) * nq);
    f->P = (double *)malloc(sizeof(double) * N * N);
    memcpy(f->mean, mean, sizeof(double) * nq);   /* Usckf.hpp:83-86 */
    memcpy(f->P, P, sizeof(double) * N * N);
    return f;
}

/* copy the 12x12 block (br,bc) <- src (12x12) inside an ld x ld matrix */
static void blk_set(double *P, int ld, int br, int bc, const double *src)
{
    for (int j = 0; j < 12; ++j)
        for (int i = 0; i < 12; ++i) AT(P, ld, 12 * br + i, 12 * bc + j) = src ? AT(src, 12, i, j) : 0.0;
}
static void blk_get(const double *P, int ld, int br, int bc, double *dst)
{
    for (int j = 0; j < 12; ++j)
        for (int i = 0; i < 12; ++i) AT(dst, 12, i, j) = AT(P, ld, 12 * br + i, 12 * bc + j);
}

/* Usckf::cloning, Usckf.hpp:391-433.  Block indices: 0 = statek, 1 = statek_l, 2 = statek_i */
void slko_usckf_cloning(slko_usckf *f, int mode)
{
    int N = slko_dof(&f->lay);
    double B[144];
    switch (mode) {
    case SLKO_STATEK_I:
        memcpy(f->mean + 13, f->mean + 26, sizeof(double) * 13);  /* statek_l = statek_i */
        blk_get(f->P, N, 2, 2, B);
        blk_set(f->P, N, 1, 1, B); blk_set(f->P, N, 1, 2, B); blk_set(f->P, N, 2, 1, B);
        blk_set(f->P, N, 0, 2, NULL); blk_set(f->P, N, 2, 0, NULL);
        blk_set(f->P, N, 0, 1, NULL); blk_set(f->P, N, 1, 0, NULL);
        break;
    case SLKO_STATEK_L:
        memcpy(f->mean, f->mean + 13, sizeof(double) * 13);       /* statek = statek_l */
        blk_get(f->P, N, 1, 1, B);
        blk_set(f->P, N, 0, 0, B); blk_set(f->P, N, 0, 1, B); blk_set(f->P, N, 1, 0, B);
        break;
    default:
        break;
    }
}

/* Usckf(single_state, P0_single), Usckf.hpp:90-103 */
slko_usckf *slko_usckf_new_single(const double *state13, const double *P0_12)
{
    slko_usckf *f = (slko_usckf *)calloc(1, sizeof(*f));
    f->lay = (slko_layout){SLKO_AUGMENTED, 0, 0, 0};
    f->mean = (double *)malloc(sizeof(double) * 39);
    f->P = (double *)calloc(36 * 36, sizeof(double));
    identity_state(&f->lay, f->mean);                 /* default-constructed AugmentedState */
    memcpy(f->mean + 26, state13, sizeof(double) * 13);
    blk_set(f->P, 36, 2, 2, P0_12);
    slko_usckf_cloning(f, SLKO_STATEK_I);
    slko_usckf_cloning(f, SLKO_STATEK_L);
    return f;
}

void slko_usckf_free(slko_usckf *f)
{
    if (!f) return;
    free(f->mean); free(f->P); free(f);
}

/* Usckf::setMeasurement, Usckf.hpp:322-389: (re)place one feature block with
 * covariance R, keep the other feature block's own covariance, wipe every
 * state<->feature cross term, restore the 36x36 state block. */
void slko_usckf_set_measurement(slko_usckf *f, int mode, const double *z, int n, const double *R)
{
    int oldN = slko_dof(&f->lay);
    int nfk = f->lay.nfk, nfkl = f->lay.nfkl;
    double states[36 * 36];
    for (int j = 0; j < 36; ++j)
        for (int i = 0; i < 36; ++i) AT(states, 36, i, j) = AT(f->P, oldN, i, j);
    if (mode != SLKO_STATEK && mode != SLKO_STATEK_L) return;

    int new_nfk = (mode == SLKO_STATEK) ? n : nfk;
    int new_nfkl = (mode == SLKO_STATEK_L) ? n : nfkl;
    int newN = 36 + new_nfk + new_nfkl;
    double *newP = (double *)calloc((size_t)newN * newN, sizeof(double));
    double *newmean = (double *)malloc(sizeof(double) * (39 + new_nfk + new_nfkl));
    memcpy(newmean, f->mean, sizeof(double) * 39);
    if (mode == SLKO_STATEK) {
        /* NB the reference reads the kept block at offset DOF + NEW |featuresk| of the OLD
         * matrix (:342, featuresk was already overwritten at :335). */
        for (int i = 0; i < n; ++i) newmean[39 + i] = z[i];
        for (int i = 0; i < nfkl; ++i) newmean[39 + n + i] = f->mean[39 + nfk + i];
        for (int j = 0; j < nfkl; ++j)
            for (int i = 0; i < nfkl; ++i) {
                int oi = 36 + new_nfk + i, oj = 36 + new_nfk + j;
                double v = (oi < oldN && oj < oldN) ? AT(f->P, oldN, oi, oj) : 0.0;
                AT(newP, newN, 36 + n + i, 36 + n + j) = v;
            }
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) AT(newP, newN, 36 + i, 36 + j) = AT(R, n, i, j);
    } else {
        for (int i = 0; i < nfk; ++i) newmean[39 + i] = f->mean[39 + i];
        for (int i = 0; i < n; ++i) newmean[39 + nfk + i] = z[i];
        for (int j = 0; j < nfk; ++j)
            for (int i = 0; i < nfk; ++i) AT(newP, newN, 36 + i, 36 + j) = AT(f->P, oldN, 36 + i, 36 + j);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) AT(newP, newN, 36 + nfk + i, 36 + nfk + j) = AT(R, n, i, j);
    }
    for (int j = 0; j < 36; ++j)
        for (int i = 0; i < 36; ++i) AT(newP, newN, i, j) = AT(states, 36, i, j);   /* :388 */
    free(f->P); free(f->mean);
    f->P = newP; f->mean = newmean;
    usckf_resize(f, new_nfk, new_nfkl);
}

/* Usckf::predict, Usckf.hpp:107-244 */
int slko_usckf_predict(slko_usckf *f, slko_process_fn fn, void *ctx, const double *Q)
{
    int N = slko_dof(&f->lay);
    int nfk = f->lay.nfk, nfkl = f->lay.nfkl;
    double Pk_i[144], Fk[144], B[144], T[144];
    blk_get(f->P, N, 2, 2, Pk_i);
    int status = predict_single(f->mean + 26, Pk_i, fn, ctx, Q, Fk, &f->mean_iters);
    blk_set(f->P, N, 2, 2, Pk_i);                                       /* :181 */
    /* clone cross blocks :191-208 */
    blk_get(f->P, N, 0, 2, B); matmul(12, 12, 12, B, 12, 0, Fk, 12, 1, T, 12); blk_set(f->P, N, 0, 2, T);
    blk_get(f->P, N, 1, 2, B); matmul(12, 12, 12, B, 12, 0, Fk, 12, 1, T, 12); blk_set(f->P, N, 1, 2, T);
    blk_get(f->P, N, 2, 0, B); matmul(12, 12, 12, Fk, 12, 0, B, 12, 0, T, 12); blk_set(f->P, N, 2, 0, T);
    blk_get(f->P, N, 2, 1, B); matmul(12, 12, 12, Fk, 12, 0, B, 12, 0, T, 12); blk_set(f->P, N, 2, 1, T);
    /* feature cross blocks :217-235 (rows 24..35 x feature columns, then the transposes) */
    int nf = nfk + nfkl;
    if (nf > 0) {
        double *Pz = (double *)malloc(sizeof(double) * 12 * nf);
        double *FP = (double *)malloc(sizeof(double) * 12 * nf);
        for (int j = 0; j < nf; ++j)
            for (int i = 0; i < 12; ++i) AT(Pz, 12, i, j) = AT(f->P, N, 24 + i, 36 + j);
        matmul(12, nf, 12, Fk, 12, 0, Pz, 12, 0, FP, 12);
        for (int j = 0; j < nf; ++j)
            for (int i = 0; i < 12; ++i) {
                AT(f->P, N, 24 + i, 36 + j) = AT(FP, 12, i, j);
                AT(f->P, N, 36 + j, 24 + i) = AT(FP, 12, i, j);
            }
        free(Pz); free(FP);
    }
    return status;
}

/* MtkMultiStateWrap::operator+(self) -> AugmentedState::boxplus(AugmentedState&)
 * (MtkWrap.hpp:277-292, State.hpp:595-611): a [+] vectorize(b), features added. */
static void aug_plus_state(const slko_layout *lay, const double *a, const double *b, double *out)
{
    int N = slko_dof(lay);
    double *v = (double *)malloc(sizeof(double) * N);
    slko_vectorize(lay, b, v);
    slko_boxplus(lay, a, v, out);
    free(v);
}

/* MtkMultiStateWrap::operator-(self) -> AugmentedState::boxminus(res, oth)
 * (MtkWrap.hpp:297-310, State.hpp:613-634): returns the STATE set(a [-] b). */
static void aug_minus_state(const slko_layout *lay, const double *a, const double *b, double *out)
{
    int N = slko_dof(lay);
    double *v = (double *)malloc(sizeof(double) * N);
    slko_boxminus(lay, a, b, v);
    slko_set_from_vector(lay, v, out);
    free(v);
}

/* Usckf::generateSigmaPoints (augmented), Usckf.hpp:532-561 */
static int usckf_sigma_points(const slko_layout *lay, const double *mu, const double *delta, const double *P, double *X)
{
    int N = slko_dof(lay), nq = slko_storage(lay);
    double *L = (double *)malloc(sizeof(double) * N * N);
    double *delta_state = (double *)malloc(sizeof(double) * nq);
    double *l_state = (double *)malloc(sizeof(double) * nq);
    double *tmp = (double *)malloc(sizeof(double) * nq);
    int fail = slko_cholesky_lower(N, P, L);
    slko_set_from_vector(lay, delta, delta_state);                 /* delta_state.set(delta) */
    aug_plus_state(lay, mu, delta_state, X);                        /* X[0] = mu + delta_state */
    for (int j = 0; j < N; ++j) {
        slko_set_from_vector(lay, L + (size_t)j * N, l_state);     /* l_state.set(L.col(j)) */
        aug_plus_state(lay, delta_state, l_state, tmp);             /* delta_state + l_state */
        aug_plus_state(lay, mu, tmp, X + (size_t)(2 * j + 1) * nq);
        aug_minus_state(lay, delta_state, l_state, tmp);            /* delta_state - l_state */
        aug_plus_state(lay, mu, tmp, X + (size_t)(2 * j + 2) * nq);
    }
    free(L); free(delta_state); free(l_state); free(tmp);
    return fail >= 0 ? SLKO_LLT_FAIL : SLKO_OK;
}

/* Usckf::update, Usckf.hpp:246-308.  gate_dof == 0: accept_any_mahalanobis_distance
 * (the default, :249); else the chi-square gate with that dof.  The unconditional
 * stdout print at :298 is dropped. */
int slko_usckf_update(slko_usckf *f, const double *z, int m, slko_measure_fn h, void *ctx,
                      const double *R, int gate_dof, int *accepted)
{
    const slko_layout *lay = &f->lay;
    int N = slko_dof(lay), nq = slko_storage(lay), S = 2 * N + 1;
    int status = SLKO_OK;
    double *X = (double *)malloc(sizeof(double) * (size_t)S * nq);
    double *Z = (double *)malloc(sizeof(double) * (size_t)S * m);
    double *zbar = (double *)malloc(sizeof(double) * m);
    double *innov = (double *)malloc(sizeof(double) * m);
    double *Sm = (double *)malloc(sizeof(double) * m * m);
    double *Sinv = (double *)malloc(sizeof(double) * m * m);
    double *covXZ = (double *)calloc((size_t)N * m, sizeof(double));
    double *K = (double *)malloc(sizeof(double) * N * m);
    double *KS = (double *)malloc(sizeof(double) * N * m);
    double *KSKt = (double *)malloc(sizeof(double) * N * N);
    double *d = (double *)malloc(sizeof(double) * N);
    double *tmp = (double *)malloc(sizeof(double) * nq);
    double *zero = (double *)calloc(N, sizeof(double));

    status |= usckf_sigma_points(lay, f->mean, zero, f->P, X);          /* :272-275 */
    for (int p = 0; p < S; ++p) h(lay, X + (size_t)p * nq, m, Z + (size_t)p * m, ctx); /* :277-278 */
    mean_vectors(Z, m, S, zbar);                                         /* :280 */
    cov_vectors(zbar, Z, m, S, Sm);
    for (int i = 0; i < m * m; ++i) Sm[i] += R[i];                       /* :282 */
    for (int p = 0; p < S; ++p) {                                        /* :283 -> :714-737 */
        aug_minus_state(lay, X + (size_t)p * nq, f->mean, tmp);          /* _State tempXi(*Xi - meanX) */
        slko_vectorize(lay, tmp, d);                                     /* tempXi.getVectorizedState() */
        for (int j = 0; j < m; ++j) {
            double dz = Z[(size_t)p * m + j] - zbar[j];
            for (int i = 0; i < N; ++i) AT(covXZ, N, i, j) += d[i] * dz;
        }
    }
    for (int i = 0; i < N * m; ++i) covXZ[i] = 0.5 * covXZ[i];
    if (slko_inverse(m, Sm, Sinv)) status |= SLKO_SINGULAR;              /* :285-286 */
    matmul(N, m, m, covXZ, N, 0, Sinv, m, 0, K, N);                      /* :288 */
    for (int r = 0; r < m; ++r) innov[r] = z[r] - zbar[r];               /* :290 */
    double d2 = 0;                                                       /* :292 */
    for (int i = 0; i < m; ++i) {
        double s = 0;
        for (int j = 0; j < m; ++j) s += AT(Sinv, m, i, j) * innov[j];
        d2 += innov[i] * s;
    }
    int ok = gate_dof ? slko_accept_mahalanobis(d2, gate_dof) : 1;
    if (ok) {                                                            /* :294-302 */
        matmul(N, m, m, K, N, 0, Sm, m, 0, KS, N);
        matmul(N, N, m, KS, N, 0, K, N, 1, KSKt, N);
        for (int i = 0; i < N * N; ++i) f->P[i] -= KSKt[i];
        for (int i = 0; i < N; ++i) {
            double s = 0;
            for (int j = 0; j < m; ++j) s += AT(K, N, i, j) * innov[j];
            d[i] = s;
        }
        slko_set_from_vector(lay, d, tmp);                               /* innovation_state.set(K*innovation) */
        double *newmean = (double *)malloc(sizeof(double) * nq);
        aug_plus_state(lay, f->mean, tmp, newmean);                      /* mu_state + innovation_state */
        memcpy(f->mean, newmean, sizeof(double) * nq);
        free(newmean);
    }
    if (accepted) *accepted = ok;
    free(X); free(Z); free(zbar); free(innov); free(Sm); free(Sinv); free(covXZ); free(K); free(KS);
    free(KSKt); free(d); free(tmp); free(zero);
    return status;
}

/* ====================================================================== */
/* batch driver for the CPU baseline                                       */
/* ====================================================================== */

int slko_msckf_step_batch(int B, int k, int m, int steps, double *mean, double *P,
                          const double *u, const double *feat, const double *z,
                          const double *Q, const double *R, int gate, unsigned *outliers)
{
    slko_layout lay = {SLKO_MULTI, k, 0, 0};
    int N = slko_dof(&lay), nq = slko_storage(&lay);
    int status = 0;
    for (int b = 0; b < B; ++b) {
        slko_msckf f;
        memset(&f, 0, sizeof(f));
        f.lay = lay;
        f.mean = mean + (size_t)b * nq;
        f.P = P + (size_t)b * N * N;
        slko_delta_pose dp;
        memcpy(dp.dpos, u + (size_t)b * 13, sizeof(double) * 3);
        memcpy(dp.dquat, u + (size_t)b * 13 + 3, sizeof(double) * 4);
        memcpy(dp.velocity, u + (size_t)b * 13 + 7, sizeof(double) * 3);
        memcpy(dp.angular_velocity, u + (size_t)b * 13 + 10, sizeof(double) * 3);
        unsigned total = 0;
        for (int s = 0; s < steps; ++s) {
            unsigned no = 0;
            status |= slko_msckf_predict(&f, slko_pm_delta_pose, &dp, Q);
            status |= slko_msckf_update(&f, z + (size_t)b * m, m, slko_mm_feature_proj,
                                        (void *)(feat + (size_t)b * (m / 2) * 4), R, gate, &no);
            total += no;
        }
        if (outliers) outliers[b] = total;
    }
    return status;
}

/* B independent Usckf filters (N = 36 + nfk + nfkl), `steps` x (predict with the constant-velocity model of
 * test/UsckfUnitTest.cpp:34-49 + update with the relative-transform model :62-86, no gate as in Usckf.hpp:249).
 * u [B][7] = velocity[3] angular_velocity[3] dt; z [B][nfk].  CPU-baseline timing of bench.py --filter usckf. */
int slko_usckf_step_batch(int B, int nfk, int nfkl, int steps, double *mean, double *P,
                          const double *u, const double *z, const double *Q, const double *R)
{
    slko_layout lay = {SLKO_AUGMENTED, 0, nfk, nfkl};
    int N = slko_dof(&lay), nq = slko_storage(&lay);
    int status = 0;
    for (int b = 0; b < B; ++b) {
        slko_usckf f;
        memset(&f, 0, sizeof(f));
        f.lay = lay;
        f.mean = mean + (size_t)b * nq;
        f.P = P + (size_t)b * N * N;
        slko_const_velocity cv;
        memcpy(cv.velocity, u + (size_t)b * 7, sizeof(double) * 3);
        memcpy(cv.angular_velocity, u + (size_t)b * 7 + 3, sizeof(double) * 3);
        cv.dt = u[(size_t)b * 7 + 6];
        for (int s = 0; s < steps; ++s) {
            int acc = 0;
            status |= slko_usckf_predict(&f, slko_pm_const_velocity, &cv, Q);
            status |= slko_usckf_update(&f, z + (size_t)b * nfk, nfk, slko_mm_vo_relative, NULL, R, 0, &acc);
        }
    }
    return status;
}

/* ====================================================================== */
/* TransformWithUncertainty (src/core/Transform.cpp) and the pose legs of */
/* DeadReckon::updatePose (src/core/DeadReckon.hpp:129-239, :306-330)     */
/* ====================================================================== */
/* A transform is stored as pos[3] quat[4: x,y,z,w]; the reference keeps an Eigen::Affine3d and derives quaternions
 * from its rotation matrix (Eigen::Quaterniond(linear()), Transform.cpp:222-225), which is restated here (Eigen
 * Quaternion.h, quaternionbase_assign_impl<Other,3,3>).  6x6 covariances are column-major in the [r t] order of
 * the reference (rotation as a scaled axis first, translation second; Transform.hpp:57-61).  3x3 helpers are
 * ROW-major m[3*i + j] internally. */

static void quat_to_rot(const double q[4], double R[9])            /* Eigen QuaternionBase::toRotationMatrix */
{
    double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

static void rot_to_quat(const double m[9], double q[4])            /* Eigen: rotation matrix -> quaternion */
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[3 * 2 + 1] - m[3 * 1 + 2]) * t;
        q[1] = (m[3 * 0 + 2] - m[3 * 2 + 0]) * t;
        q[2] = (m[3 * 1 + 0] - m[3 * 0 + 1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[3 * i + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[3 * i + i] - m[3 * j + j] - m[3 * k + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}

/* q_to_r, Transform.cpp:44-48: Eigen::AngleAxisd(q) (Eigen 3.3: angle = 2 atan2(|vec|, |w|), axis = vec / (+-|vec|)) */
static void q_to_r(const double q[4], double r[3])
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n != 0) {
        double angle = 2 * atan2(n, fabs(q[3]));
        if (q[3] < 0) n = -n;
        r[0] = q[0] / n * angle; r[1] = q[1] / n * angle; r[2] = q[2] / n * angle;
    } else { r[0] = r[1] = r[2] = 0; }                              /* angle 0 times the axis (1, 0, 0) */
}

static double sign_of(double v) { return v > 0 ? 1.0 : -1.0; }     /* Transform.cpp:50-53 */

static void skew(const double r[3], double S[9])                  /* Transform.cpp:55-62 */
{
    S[0] = 0; S[1] = -r[2]; S[2] = r[1];
    S[3] = r[2]; S[4] = 0; S[5] = -r[0];
    S[6] = -r[1]; S[7] = r[0]; S[8] = 0;
}

static void mat_mul(int n, int k, int m, const double *A, const double *B, double *C)   /* row-major (n x k)(k x m) */
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0;
            for (int p = 0; p < k; ++p) s += A[i * k + p] * B[p * m + j];
            C[i * m + j] = s;
        }
}

/* dq_by_dr, Transform.cpp:64-76: 4 x 3, quaternion ordered (w, x, y, z) */
static void dq_by_dr(const double q[4], double D[12])
{
    double r[3];
    q_to_r(q, r);
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double kappa = 0.5 - theta * theta / 48.0;
    double lambda = 1.0 / 24.0 * (1.0 - theta * theta / 40.0);
    for (int j = 0; j < 3; ++j) D[j] = -q[j] / 2.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[3 * (i + 1) + j] = kappa * (i == j) - lambda * r[i] * r[j];
}

/* dr_by_dq, Transform.cpp:78-89: 3 x 4 */
static void dr_by_dq(const double q[4], double D[12])
{
    double mu = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    double tau = 2.0 * sign_of(q[3]) * (1.0 + mu * mu / 6.0);
    double nu = -2.0 * sign_of(q[3]) * (2.0 / 3.0 + mu * mu / 5.0);
    for (int i = 0; i < 3; ++i) {
        D[4 * i] = -2 * q[i];
        for (int j = 0; j < 3; ++j) D[4 * i + 1 + j] = tau * (i == j) + nu * q[i] * q[j];
    }
}

/* dq2q1_by_dq1(q2) (sgn = +1) and dq2q1_by_dq2(q1) (sgn = -1), Transform.cpp:91-105: 4 x 4 */
static void dq2q1_by(const double q[4], double sgn, double M[16])
{
    double S[9];
    skew(q, S);
    for (int i = 0; i < 16; ++i) M[i] = 0;
    for (int j = 0; j < 3; ++j) { M[1 + j] = -q[j]; M[4 * (1 + j)] = q[j]; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[4 * (1 + i) + 1 + j] = sgn * S[3 * i + j];
    for (int i = 0; i < 4; ++i) M[5 * i] += q[3];
}

/* dr2r1_by_r1 / dr2r1_by_r2, Transform.cpp:107-121 */
static void dr2r1_by(const double q[4], const double qa[4] /* 4x4 from */, double sgn, const double qb[4] /* dq_by_dr of */, double J[9])
{
    double A[12], M[16], B[12], T[12];
    dr_by_dq(q, A);
    dq2q1_by(qa, sgn, M);
    dq_by_dr(qb, B);
    mat_mul(3, 4, 4, A, M, T);
    mat_mul(3, 4, 3, T, B, J);
}

/* drx_by_dr, Transform.cpp:123-137 */
static void drx_by_dr(const double q[4], const double x[3], double J[9])
{
    double r[3];
    q_to_r(q, r);
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double alpha = 1.0 - theta * theta / 6.0, beta = 0.5 - theta * theta / 24.0;
    double gamma = 1.0 / 3.0 - theta * theta / 30.0, delta = -1.0 / 12.0 + theta * theta / 180.0;
    double Sx[9], Sr[9], A[9], B[9], T1[9], T2[9], T3[9];
    skew(x, Sx);
    skew(r, Sr);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[3 * i + j] = gamma * r[i] * r[j] - beta * Sr[3 * i + j] + alpha * (i == j);
            B[3 * i + j] = delta * r[i] * r[j] + 2.0 * beta * (i == j);
        }
    mat_mul(3, 3, 3, Sx, A, T1);
    mat_mul(3, 3, 3, Sr, Sx, T2);
    mat_mul(3, 3, 3, T2, B, T3);
    for (int i = 0; i < 9; ++i) J[i] = -T1[i] - T3[i];
}

/* cov (6x6 column-major) += J C J^T with J given as four row-major 3x3 blocks [[J00, J01], [J10, J11]] */
static void add_jcjt(const double *J00, const double *J01, const double *J10, const double *J11, const double *C, double *cov)
{
    double J[36], T[36];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            J[6 * i + j] = J00[3 * i + j]; J[6 * i + 3 + j] = J01[3 * i + j];
            J[6 * (3 + i) + j] = J10[3 * i + j]; J[6 * (3 + i) + 3 + j] = J11[3 * i + j];
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int p = 0; p < 6; ++p) s += J[6 * i + p] * C[p + 6 * j];          /* C column-major */
            T[6 * i + j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int p = 0; p < 6; ++p) s += T[6 * i + p] * J[6 * j + p];
            cov[i + 6 * j] += s;
        }
}

/* TransformWithUncertainty::operator* (Transform.cpp:215-254): result = t2 * t1.  cov1 / cov2 NULL = that transform
 * carries no uncertainty (hasUncertainty() false); with both NULL the result has none either (out_cov zeroed). */
void slko_transform_compose(const double t2[7], const double *cov2, const double t1[7], const double *cov1,
                            double out_t[7], double out_cov[36])
{
    double R1[9], R2[9], R[9], q1[4], q2[4], q[4], Z[9] = {0}, I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    quat_to_rot(t1 + 3, R1);
    quat_to_rot(t2 + 3, R2);
    mat_mul(3, 3, 3, R2, R1, R);
    for (int i = 0; i < 3; ++i) out_t[i] = R2[3 * i] * t1[0] + R2[3 * i + 1] * t1[1] + R2[3 * i + 2] * t1[2] + t2[i];
    rot_to_quat(R, out_t + 3);
    for (int i = 0; i < 36; ++i) out_cov[i] = 0;
    if (!cov1 && !cov2) return;                                                     /* :219-220 */
    rot_to_quat(R1, q1);                                                            /* :222-225 */
    rot_to_quat(R2, q2);
    slko_quat_mul(q2, q1, q);
    if (cov1) {                                                                     /* :232-239 */
        double J00[9];
        dr2r1_by(q, q2, 1.0, q1, J00);
        add_jcjt(J00, Z, Z, R2, cov1, out_cov);
    }
    if (cov2) {                                                                     /* :241-248 */
        double J00[9], J10[9];
        dr2r1_by(q, q1, -1.0, q2, J00);
        drx_by_dr(q2, t1, J10);
        add_jcjt(J00, Z, J10, I, cov2, out_cov);
    }
}

/* DeadReckon::updatePose, Affine3d overload (src/core/DeadReckon.hpp:306-330): composition with uncertainty, or
 * post = prev * delta with postCov = prevCov + deltaCov.  base::guaranteeSPD(postCov) (:326) returns the repaired
 * matrix, which the reference discards: no effect. */
void slko_update_pose_affine(const double prev[7], const double prev_cov[36], const double delta[7], const double delta_cov[36],
                             int use_tf, double post[7], double post_cov[36])
{
    if (use_tf) { slko_transform_compose(prev, prev_cov, delta, delta_cov, post, post_cov); return; }
    slko_transform_compose(prev, NULL, delta, NULL, post, post_cov);
    for (int i = 0; i < 36; ++i) post_cov[i] = prev_cov[i] + delta_cov[i];
}

/* DeadReckon::updatePose, RigidBodyState overload (src/core/DeadReckon.hpp:129-239).
 *   u       = dt v0[3] w0[3] v1[3] w1[3]           (cartesianVelocities[0] = current, [1] = previous sample)
 *   velcov  = 6x6 column-major, linear 0-2 / angular 3-5; any NaN entry -> the delta covariances are zero (:165-176)
 *   pose records: pos[3] quat[4] cov_position[9] cov_orientation[9] (3x3 column-major)          = 25
 *   post (in/out) = pose record + velocity[3] cov_velocity[9] angular_velocity[3] cov_angular_velocity[9] = 49
 *   delta (out)   = pose record + velocity[3] angular_velocity[3]                                 = 31 */
void slko_dead_reckon_pose(const double u[13], const double velcov[36], const double prev[25], double post[49],
                           double delta[31], int use_tf)
{
    const double dt = u[0];
    double d13[13];
    slko_dead_reckon_delta(u, d13);                                   /* position, orientation (updateAttitude), velocities */
    for (int i = 0; i < 7; ++i) delta[i] = d13[i];
    for (int i = 0; i < 3; ++i) { delta[25 + i] = d13[7 + i]; delta[28 + i] = d13[10 + i]; }
    int has_nan = 0;
    for (int i = 0; i < 36; ++i) if (velcov[i] != velcov[i]) has_nan = 1;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            delta[7 + i + 3 * j] = has_nan ? 0.0 : velcov[i + 6 * j] * dt * dt;                      /* :171 */
            delta[16 + i + 3 * j] = has_nan ? 0.0 : velcov[(3 + i) + 6 * (3 + j)] * dt * dt;         /* :172 */
        }
    if (use_tf) {                                                     /* :202-215 */
        double c2[36] = {0}, c1[36] = {0}, t[7], c[36];
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {                             /* Transform.cpp:294-296: [cov_orientation 0; 0 cov_position] */
                c2[i + 6 * j] = prev[16 + i + 3 * j]; c2[(3 + i) + 6 * (3 + j)] = prev[7 + i + 3 * j];
                c1[i + 6 * j] = delta[16 + i + 3 * j]; c1[(3 + i) + 6 * (3 + j)] = delta[7 + i + 3 * j];
            }
        slko_transform_compose(prev, c2, delta, c1, t, c);
        for (int i = 0; i < 7; ++i) post[i] = t[i];
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {                             /* copyToRigidBodyState, Transform.cpp:314-321 */
                post[16 + i + 3 * j] = c[i + 6 * j];
                post[7 + i + 3 * j] = c[(3 + i) + 6 * (3 + j)];
            }
    } else {                                                          /* :216-223 */
        double rp[3], q[4];
        slko_quat_rotate(prev + 3, delta, rp);
        for (int i = 0; i < 3; ++i) post[i] += rp[i];
        for (int i = 0; i < 9; ++i) { post[7 + i] += delta[7 + i]; post[16 + i] += delta[16 + i]; }
        slko_quat_mul(prev + 3, delta + 3, q);
        for (int i = 0; i < 4; ++i) post[3 + i] = q[i];
    }
    for (int i = 0; i < 3; ++i) { post[25 + i] = u[1 + i]; post[37 + i] = u[4 + i]; }            /* :226-229 */
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            post[28 + i + 3 * j] = velcov[i + 6 * j];
            post[40 + i + 3 * j] = velcov[(3 + i) + 6 * (3 + j)];
        }
}

/* ====================================================================== */
/* AdaptiveAttitudeCov::matrix (src/filters/MeasurementModels.hpp:181-286) */
/* ====================================================================== */
/* Singular value decomposition of the symmetric positive semi-definite 3x3 Uk (a mean of outer products): the
 * reference calls Eigen::JacobiSVD<MatrixXd>(Uk, ComputeThinU) -- singular values sorted in decreasing order, U =
 * left singular vectors.  For a symmetric PSD matrix these are its eigenvalues / eigenvectors; restated as a cyclic
 * Jacobi eigenvalue iteration (U is used only through u u^T and u^T M u, so column signs do not matter). */
static void sym3_svd(const double A[9] /* row-major symmetric */, double s[3], double U[9] /* columns = vectors, row-major */)
{
    double a[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) a[i] = A[i];
    for (int sweep = 0; sweep < 12; ++sweep) {
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[3 * p + q];
                if (apq == 0.0) continue;
                double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k) {                        /* A <- A G */
                    double akp = a[3 * k + p], akq = a[3 * k + q];
                    a[3 * k + p] = c * akp - sn * akq;
                    a[3 * k + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {                        /* A <- G^T A */
                    double apk = a[3 * p + k], aqk = a[3 * q + k];
                    a[3 * p + k] = c * apk - sn * aqk;
                    a[3 * q + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - sn * vkq;
                    V[3 * k + q] = sn * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    double e[3] = {fabs(a[0]), fabs(a[4]), fabs(a[8])};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (e[idx[j]] < e[idx[j + 1]]) { int t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t; }
    for (int c = 0; c < 3; ++c) {
        s[c] = e[idx[c]];
        for (int k = 0; k < 3; ++k) U[3 * k + c] = V[3 * k + idx[c]];
    }
}

/* One call of AdaptiveAttitudeCov::matrix.  State of the object: hist [m1][9] (row-major 3x3 each, zero-initialised,
 * :162-165), *r1count (starts at 0, :160), *r2count (constructor argument R2COUNT, :158).
 * xk [n], Pk [n*n] column-major, z [3], H [3*n] column-major (3 x n), R [9] column-major; Rout [9] column-major. */
void slko_adaptive_attitude_cov(unsigned m1, unsigned m2, double gamma, double *hist, unsigned *r1count, unsigned *r2count,
                                int n, const double *xk, const double *Pk, const double *z, const double *H, const double *R,
                                double *Rout)
{
    double res[3], Uk[9] = {0}, fooR[9], s[3], U[9], mu[3], Qstar[9] = {0};
    for (int i = 0; i < 3; ++i) {                                    /* z - H xk, :195 */
        double hx = 0;
        for (int j = 0; j < n; ++j) hx += H[i + 3 * j] * xk[j];
        res[i] = z[i] - hx;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) hist[9 * (*r1count) + 3 * i + j] = res[i] * res[j];        /* :195-197 */
    *r1count = (*r1count + 1) % m1;                                  /* :213 */
    for (unsigned h = 0; h < m1; ++h)                                /* :215-223 */
        for (int i = 0; i < 9; ++i) Uk[i] += hist[9 * h + i];
    for (int i = 0; i < 9; ++i) Uk[i] = Uk[i] / (double)m1;
    for (int i = 0; i < 3; ++i)                                      /* fooR = H Pk H^T + R, :225 */
        for (int j = 0; j < 3; ++j) {
            double sum = 0;
            for (int a = 0; a < n; ++a) {
                double hp = 0;
                for (int b = 0; b < n; ++b) hp += H[i + 3 * b] * Pk[b + n * a];
                sum += hp * H[j + 3 * a];
            }
            fooR[3 * i + j] = sum + R[i + 3 * j];
        }
    sym3_svd(Uk, s, U);                                              /* :230-235 */
    for (int c = 0; c < 3; ++c) {                                    /* mu_c = u_c^T fooR u_c, :237-239 */
        double sum = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) sum += U[3 * i + c] * fooR[3 * i + j] * U[3 * j + c];
        mu[c] = sum;
    }
    double mx = s[0] - mu[0];
    for (int c = 1; c < 3; ++c) if (s[c] - mu[c] > mx) mx = s[c] - mu[c];
    int use = 0;
    if (mx > gamma) { *r2count = 0; use = 1; }                       /* :245-258 */
    else { *r2count = *r2count + 1; use = *r2count < m2; }           /* :259-275 */
    if (use)
        for (int c = 0; c < 3; ++c) {
            double w = s[c] - mu[c] > 0.0 ? s[c] - mu[c] : 0.0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Qstar[3 * i + j] += w * U[3 * i + c] * U[3 * j + c];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rout[i + 3 * j] = R[i + 3 * j] + Qstar[3 * i + j];          /* :284 */
}
