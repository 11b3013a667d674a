/*
 * slk_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A dependency-free fp64 C restatement of the sigma-point Kalman hot path of
 * jhidalgocarrio/slam-localization: localization::Usckf / localization::Msckf
 * predict() and update() (reference src/filters/Usckf.hpp, Msckf.hpp,
 * State.hpp, MtkWrap.hpp).  Every function cites the reference file:line it
 * follows.  The manifold arithmetic of the reference lives in third-party MTK
 * (Rock package slam/mtk, manifest.xml:16, un-versioned) which is NOT present
 * under /root/reference; its published SO(3) algorithm is restated in
 * slk_oracle.c (see the comment on so3_exp / so3_log).
 *
 * PARITY STATUS: "parity unpinned" for filter outputs -- the reference holds no
 * golden vectors, tolerances or known answers for predict()/update()
 * (test/UsckfUnitTest.cpp and test/MsckfUnitTest.cpp only print), and the
 * reference cannot be compiled here (Eigen, Boost, MTK, ukfom, base-types are
 * absent).  What IS pinned: the manifold identities the reference asserts
 * (test/MsckfUnitTest.cpp:61,62,66,71,110,113), closed-form Kalman known-answer
 * tests, the checkSigmaPoints invariant (Usckf.hpp:769-789) and an independent
 * numpy/scipy re-implementation (oracle/np_check.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in this directory.
 *
 * Conventions: all matrices are column-major (Eigen default).  Quaternions are
 * stored (x, y, z, w) like Eigen::Quaternion::coeffs().
 *   State   storage (13): pos[3] quat[4] velo[3] angvelo[3]   tangent DOF 12
 *   Sensor  storage  (7): pos[3] quat[4]                      tangent DOF 6
 *   MultiState     = State + k * Sensor                        (State.hpp:336-527)
 *   AugmentedState = statek, statek_l, statek_i, featuresk[nfk], featuresk_l[nfkl]
 *                                                              (State.hpp:529-669)
 */
#ifndef SLK_ORACLE_H
#define SLK_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { SLKO_SINGLE = 0, SLKO_MULTI = 1, SLKO_AUGMENTED = 2 };
enum { SLKO_STATEK = 1, SLKO_STATEK_L = 2, SLKO_STATEK_I = 3 }; /* Usckf.hpp:37-42 */

/* status bits returned by the filter calls (the reference has none: it asserts
 * or silently continues; Usckf.hpp:537-538, 620-624) */
enum {
    SLKO_OK = 0,
    SLKO_LLT_FAIL = 1,        /* non-positive pivot in a Cholesky factorisation */
    SLKO_MEAN_NOT_CONVERGED = 2,
    SLKO_SINGULAR = 4,        /* zero pivot in an LU inverse */
    SLKO_EKF_ROWS = 16        /* EKF update: fewer measurement rows than state dimensions survive the gate */
};

typedef struct {
    int kind;   /* SLKO_SINGLE / SLKO_MULTI / SLKO_AUGMENTED */
    int k;      /* number of sensor-pose clones (SLKO_MULTI) */
    int nfk;    /* |featuresk|   (SLKO_AUGMENTED) */
    int nfkl;   /* |featuresk_l| (SLKO_AUGMENTED) */
} slko_layout;

int slko_dof(const slko_layout *lay);      /* tangent dimension N   */
int slko_storage(const slko_layout *lay);  /* stored mean length Nq */

/* ---- SO(3) / manifold primitives ------------------------------------- */
void slko_so3_exp(const double v[3], double scale, double q[4]);
void slko_so3_log(const double q[4], double v[3]);
void slko_quat_mul(const double a[4], const double b[4], double out[4]);
void slko_quat_rotate(const double q[4], const double v[3], double out[3]);
void slko_boxplus(const slko_layout *lay, const double *x, const double *v, double *out);
void slko_boxminus(const slko_layout *lay, const double *a, const double *b, double *out);
void slko_set_from_vector(const slko_layout *lay, const double *v, double *x);
void slko_vectorize(const slko_layout *lay, const double *x, double *v);

/* ---- dense helpers ---------------------------------------------------- */
int slko_cholesky_lower(int n, const double *A, double *L);  /* returns -1 ok, else failing pivot */
int slko_inverse(int n, const double *A, double *Ainv);      /* partial-pivot LU inverse; 0 ok */

/* ---- model callbacks -------------------------------------------------- */
typedef void (*slko_process_fn)(const double *x13, double *y13, void *ctx);
typedef void (*slko_measure_fn)(const slko_layout *lay, const double *X, int m, double *z, void *ctx);

/* process models of the reference tests */
typedef struct { double velocity[3], angular_velocity[3], dt; } slko_const_velocity;       /* UsckfUnitTest.cpp:34-49 */
typedef struct { double dpos[3], dquat[4], velocity[3], angular_velocity[3]; } slko_delta_pose; /* MsckfUnitTest.cpp:33-47 */
void slko_pm_const_velocity(const double *x, double *y, void *ctx);
void slko_pm_delta_pose(const double *x, double *y, void *ctx);
/* dead reckoning (src/core/DeadReckon.hpp:129-239, :246-286): u = dt v0[3] w0[3] v1[3] w1[3] */
void slko_update_attitude(double dt, const double w0[3], const double w1[3], double q[4]);
void slko_dead_reckon_delta(const double u[13], double delta[13]);
void slko_pm_dead_reckon(const double *x, double *y, void *ctx);   /* ctx = double[13] u */

/* TransformWithUncertainty::operator* (src/core/Transform.cpp:215-254, Jacobians :35-137) on pos[3] quat[4] transforms
 * with 6x6 [r t] covariances (NULL = no uncertainty), the Affine3d and RigidBodyState overloads of
 * DeadReckon::updatePose (src/core/DeadReckon.hpp:306-330, :129-239) and AdaptiveAttitudeCov::matrix
 * (src/filters/MeasurementModels.hpp:181-286).  Record layouts: see slk_oracle.c. */
void slko_transform_compose(const double t2[7], const double *cov2, const double t1[7], const double *cov1,
                            double out_t[7], double out_cov[36]);
void slko_update_pose_affine(const double prev[7], const double prev_cov[36], const double delta[7], const double delta_cov[36],
                             int use_tf, double post[7], double post_cov[36]);
void slko_dead_reckon_pose(const double u[13], const double velcov[36], const double prev[25], double post[49],
                           double delta[31], int use_tf);
void slko_adaptive_attitude_cov(unsigned m1, unsigned m2, double gamma, double *hist, unsigned *r1count, unsigned *r2count,
                                int n, const double *xk, const double *Pk, const double *z, const double *H, const double *R,
                                double *Rout);

/* measurement models */
void slko_mm_vo_relative(const slko_layout *lay, const double *X, int m, double *z, void *ctx); /* UsckfUnitTest.cpp:62-86 */
/* ctx = double[ (m/2) * 4 ]: per 2-D feature (landmark xyz, pose index); pose 0 = statek, 1.. = clones */
void slko_mm_feature_proj(const slko_layout *lay, const double *X, int m, double *z, void *ctx);
/* ctx = double[1] pose index; z = position of that pose (m = 3) */
void slko_mm_pose_position(const slko_layout *lay, const double *X, int m, double *z, void *ctx);

/* ---- Msckf ------------------------------------------------------------ */
typedef struct {
    slko_layout lay;
    double *mean;  /* Nq */
    double *P;     /* N x N col-major */
    double Fk[144];   /* last predict's Fk (Msckf.hpp:138), 12x12 col-major */
    int mean_iters;   /* trip count of the last manifold mean */
} slko_msckf;

slko_msckf *slko_msckf_new(int k, const double *mean, const double *P);
void slko_msckf_free(slko_msckf *f);
int slko_msckf_predict(slko_msckf *f, slko_process_fn fn, void *ctx, const double *Q);
/* EKF update, Msckf.hpp:284-349 (removeOutliers :756-789, reduceDimension :791-816): zmean = h(mu) and H (m x N,
 * column-major) are what the reference's functor h(mu_state, H) returns */
int slko_msckf_update_ekf(slko_msckf *f, const double *z, const double *zmean, const double *H, int m,
                          const double *R, int gate, unsigned *n_outliers);
int slko_msckf_update(slko_msckf *f, const double *z, int m, slko_measure_fn h, void *ctx,
                      const double *R, int gate, unsigned *n_outliers);
/* pieces, exposed for unit tests */
int slko_msckf_check_sigma_points(const slko_msckf *f, double *max_cov_err, double *mean_err);

/* ---- Usckf ------------------------------------------------------------ */
typedef struct {
    slko_layout lay;
    double *mean;   /* 39 + nfk + nfkl */
    double *P;      /* N x N col-major, N = 36 + nfk + nfkl */
    int mean_iters;
} slko_usckf;

slko_usckf *slko_usckf_new_single(const double *state13, const double *P0_12);
slko_usckf *slko_usckf_new(int nfk, int nfkl, const double *mean, const double *P);
void slko_usckf_free(slko_usckf *f);
void slko_usckf_cloning(slko_usckf *f, int mode);
void slko_usckf_set_measurement(slko_usckf *f, int mode, const double *z, int n, const double *R);
int slko_usckf_predict(slko_usckf *f, slko_process_fn fn, void *ctx, const double *Q);
int slko_usckf_update(slko_usckf *f, const double *z, int m, slko_measure_fn h, void *ctx,
                      const double *R, int gate_dof, int *accepted);

int slko_accept_mahalanobis(double d2, int dof);  /* Msckf.hpp:844-905 */

/* ---- batch driver (CPU baseline timing in bench.py; single thread) ---- */
/* B independent Msckf filters, `steps` x (predict with delta-pose model + update with
 * the feature-projection model).  mean [B][Nq], P [B][N*N], u [B][13] (delta-pose
 * params), feat [B][(m/2)*4], z [B][m], Q [144], R [m*m].  Returns OR of statuses. */
int slko_msckf_step_batch(int B, int k, int m, int steps, double *mean, double *P,
                          const double *u, const double *feat, const double *z,
                          const double *Q, const double *R, int gate, unsigned *outliers);

/* same for B Usckf filters: constant-velocity process model + relative-transform measurement model, u [B][7],
 * z [B][nfk], no gate */
int slko_usckf_step_batch(int B, int nfk, int nfkl, int steps, double *mean, double *P,
                          const double *u, const double *z, const double *Q, const double *R);

#ifdef __cplusplus
}
#endif
#endif
