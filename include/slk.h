/*
 * slk.h -- C ABI of the MI355X-native sigma-point Kalman hot path
 *          ("slk" = sigma-point localization kernels).
 *
 * Drop-in boundary for the predict()/update() path of
 *   localization::Msckf<_MultiState,_SingleState>      reference src/filters/Msckf.hpp
 *   localization::Usckf<_AugmentedState,_SingleState>  reference src/filters/Usckf.hpp
 * The reference is a header-only C++ template library with no FFI of its own; this is the
 * interface its filter classes bind to when they are backed by the GPU (the header facade
 * under include/localization/filters/ calls exactly these entry points; INTEGRATION.md shows
 * the binding).  One handle = a BATCH of B independent filters of identical layout on one
 * device; B = 1 reproduces the reference object.
 *
 * Conventions
 *   - fp64 everywhere (the reference scalar is double: src/filters/State.hpp:37-38).
 *   - matrices are column-major (Eigen default), one filter after the other:
 *       mean [B][Nq], P [B][N*N], Q [12*12], R [m*m].
 *   - quaternions are stored (x, y, z, w) like Eigen::Quaternion::coeffs().
 *   - State   storage (13): pos[3] quat[4] velo[3] angvelo[3], tangent DOF 12 (State.hpp:137-149)
 *     Sensor  storage  (7): pos[3] quat[4],                    tangent DOF  6 (State.hpp:242-252)
 *     Msckf mean  = State + k * Sensor            N = 12 + 6k      (State.hpp:336-376)
 *     Usckf mean  = statek, statek_l, statek_i, featuresk[nfk], featuresk_l[nfkl]
 *                                                 N = 36 + nfk + nfkl (State.hpp:529-593)
 *   - every pointer argument is host or device memory as said by the `where` argument of the
 *     call (SLK_HOST: the library stages it through its own device buffers; SLK_DEVICE: used
 *     in place, must be resident on the handle's device).
 *   - return value: 0 = ok, < 0 = API / runtime error (nothing was changed).  Numerical
 *     conditions are per-filter STATUS bits (slk_get_status); the reference has no error
 *     reporting at all (asserts / silently ignored LLT failure: Usckf.hpp:537-538, 620-624).
 *   - a handle is not thread-safe; distinct handles may be used concurrently.  All work of a
 *     handle is ordered on one HIP stream.
 */
#ifndef SLK_H
#define SLK_H

#ifdef __cplusplus
extern "C" {
#endif

#define SLK_ABI_VERSION 1

typedef struct slk_filter slk_filter;

enum { SLK_MSCKF = 1, SLK_USCKF = 2 };
enum { SLK_HOST = 0, SLK_DEVICE = 1 };

/* error codes */
enum {
    SLK_OK = 0,
    SLK_E_INVALID = -1,      /* bad argument / size mismatch (reference: assert, Usckf.hpp:325-327) */
    SLK_E_NO_DEVICE = -2,    /* no usable HIP device: the product has NO CPU fallback */
    SLK_E_HIP = -3,          /* HIP runtime error, see slk_last_error() */
    SLK_E_UNSUPPORTED = -4,  /* shape outside what the kernels are built for */
    SLK_E_NOMEM = -5
};

/* per-filter status bits (OR-accumulated until slk_clear_status) */
enum {
    SLK_ST_LLT_FAIL = 1,            /* non-positive Cholesky pivot; that call left the filter unchanged */
    SLK_ST_MEAN_NOT_CONVERGED = 2,  /* manifold mean hit max_it = 10000 (Msckf.hpp:475,489-493) */
    SLK_ST_SINGULAR = 4,            /* innovation covariance not invertible */
    SLK_ST_ALL_REJECTED = 8,        /* every measurement block failed the gate: update skipped (Msckf.hpp:250) */
    SLK_ST_EKF_ROWS = 16,           /* EKF update: fewer rows than state dimensions survive the gate; the reference would
                                       read R.block(0,0,N,N) out of range (Msckf.hpp:806): update skipped */
    SLK_ST_BAD_INDEX = 32           /* a pose index among the parameters of a registered measurement model is outside
                                       0..k (Msckf) / 0..2 (Usckf) or not a number: update skipped (device-resident
                                       parameters; host-resident ones are rejected with SLK_E_INVALID before the launch) */
};

/* cloning modes of Usckf (Usckf.hpp:37-42) */
enum { SLK_STATEK = 1, SLK_STATEK_L = 2, SLK_STATEK_I = 3 };

/* Registered process models f: SingleState -> SingleState (Tier A, run on the GPU).
 * SLK_MODEL_EXTERNAL = opaque host functor (Tier B): the caller maps the sigma points itself. */
enum {
    SLK_MODEL_EXTERNAL = 0,
    SLK_PM_CONST_VELOCITY = 1,  /* test/UsckfUnitTest.cpp:34-49; u = velocity[3] angular_velocity[3] dt  (7) */
    SLK_PM_DELTA_POSE = 2,      /* test/MsckfUnitTest.cpp:33-47; u = dpos[3] dquat[4] velocity[3] angular_velocity[3] (13) */
    SLK_PM_DEAD_RECKON = 3      /* the step before predict fused in: DeadReckon::updatePose delta pose
                                   (src/core/DeadReckon.hpp:129-239, updateAttitude :246-286) feeding the delta-pose model;
                                   u = dt, current velocity[3] angular_velocity[3], previous velocity[3] angular_velocity[3] (13) */
};

/* Registered measurement models h: FullState -> R^m */
enum {
    SLK_MM_VO_RELATIVE = 1,     /* test/UsckfUnitTest.cpp:62-86 (Usckf only); no parameters, m = nfk */
    SLK_MM_FEATURE_PROJ = 2,    /* m/2 landmarks seen as normalised image points from a pose;
                                   params = (m/2) x { landmark xyz, pose index }; pose 0 = current
                                   state, c >= 1 = clone c-1 (Msckf) / 0,1,2 = statek,statek_l,statek_i (Usckf) */
    SLK_MM_POSE_POSITION = 3    /* z = position of pose params[0]; m = 3 */
};

typedef struct {
    int kind;            /* SLK_MSCKF / SLK_USCKF */
    int batch;           /* B >= 1 independent filters */
    int device;          /* HIP device ordinal */
    int n_clones;        /* Msckf: k sensor-pose clones (MultiState::sensorsk.size()) */
    int n_featuresk;     /* Usckf: |featuresk| */
    int n_featuresk_l;   /* Usckf: |featuresk_l| */
    void *stream;        /* hipStream_t to run on, or NULL for a library-owned stream */
} slk_config;

/* ---- lifetime: replaces the filter constructors (Msckf.hpp:80-85, Usckf.hpp:83-103) ---- */
int  slk_create(const slk_config *cfg, slk_filter **out);
void slk_destroy(slk_filter *f);
const char *slk_last_error(void);
int  slk_device_count(void);

int slk_batch(const slk_filter *f);
int slk_dof(const slk_filter *f);       /* N  = getDOF()            (State.hpp:373-376, 590-593) */
int slk_storage(const slk_filter *f);   /* Nq = stored mean length */

/* ---- state access: muState()/getPk()/setPk()/muSingleState() (Msckf.hpp:351-395,
 *      Usckf.hpp:435-526).  mean [B][Nq], P [B][N*N]; either may be NULL. ---- */
int slk_set_state(slk_filter *f, const double *mean, const double *P, int where);
int slk_get_state(slk_filter *f, double *mean, double *P, int where);
double *slk_mean_device_ptr(slk_filter *f);   /* resident buffers, for zero-copy callers */
double *slk_cov_device_ptr(slk_filter *f);    /* (the exact-shape Msckf update kernels store the lower triangle and the
                                                * diagonal tiles of P+ only -- nothing on the device reads more, Msckf.hpp:412, :447 --
                                                * and this call enqueues the mirror pass that completes the strict upper triangle:
                                                * call it again after further steps before reading the upper triangle through the
                                                * pointer; slk_get_state and every other entry point do the same by themselves) */

/* ---- predict(f, Q): Msckf.hpp:89-189, Usckf.hpp:107-244.
 *      u [B][u_stride] model inputs (u_stride 0 = one shared row);
 *      Q 12x12 (q_stride 0 = shared, else per-filter stride in doubles). ---- */
int slk_predict(slk_filter *f, int model, const double *u, int u_stride,
                const double *Q, int q_stride, int where);

/* ---- DeadReckon::updatePose (src/core/DeadReckon.hpp:129-239), delta-pose part, for the whole batch:
 *      u [B][u_stride] = dt v0[3] w0[3] v1[3] w1[3]  ->  delta [B][13] = dpos[3] dquat[4] velocity[3] angular_velocity[3],
 *      which is the `u` of slk_predict(SLK_PM_DELTA_POSE).  The covariance part of the reference
 *      (cov_position = C_vv dt^2, cov_orientation = C_ww dt^2, :167-176) is a scaling left to the caller. ---- */
int slk_dead_reckon(slk_filter *f, const double *u, int u_stride, double *delta, int where);

/* ---- TransformWithUncertainty::operator* (src/core/Transform.cpp:215-254; Jacobians of Pennec & Thirion :35-137) for the
 *      batch: out = t2 * t1 with transforms [B][7] = pos[3] quat[4: x,y,z,w] and covariances [B][36] = 6x6 column-major
 *      in the reference's [r t] order (rotation as a scaled axis first, translation second; Transform.hpp:57-61).
 *      cov2 / cov1 NULL = that side carries no uncertainty (hasUncertainty() false).  additive != 0 is the other branch
 *      of DeadReckon::updatePose's Affine3d overload (src/core/DeadReckon.hpp:306-330): pose = t2 * t1, covariance =
 *      cov2 + cov1 (t2 = prevPose, t1 = deltaPose there).  cov_out may be NULL.
 *      Eigen semantics mirrored (the reference pins no Eigen version and holds no fixture for Transform -- the behaviour
 *      below is pinned by this library's own tests, parity with the reference is UNPINNED): quaternion from the rotation
 *      matrix as Eigen::Quaterniond(linear()); q_to_r (Transform.cpp:44-48) as Eigen >= 3.3's AngleAxisd(q), i.e. angle =
 *      2 atan2(|vec|, |w|) with the axis flipped for w < 0 -- Eigen 3.0 - 3.2 take 2 acos(w) with the axis as it is, which
 *      gives the other representative of the rotation vector (2 pi apart) for a composite quaternion with w < 0. ---- */
int slk_transform_compose(slk_filter *f, const double *t2, const double *cov2, const double *t1, const double *cov1,
                          double *t_out, double *cov_out, int additive, int where);

/* ---- DeadReckon::updatePose, RigidBodyState overload (src/core/DeadReckon.hpp:129-239), whole: delta pose AND the
 *      covariance / posterior legs (:165-176, :200-229), for the batch.
 *      u [B][u_stride] as in slk_dead_reckon; velcov 6x6 column-major (linear 0-2, angular 3-5), c_stride 0 = shared;
 *      an entry that is NaN zeroes the delta covariances (:165-176).
 *      prev  [B][25] = pos[3] quat[4] cov_position[9] cov_orientation[9]            (3x3 column-major)
 *      post  [B][49] = the same 25 + velocity[3] cov_velocity[9] angular_velocity[3] cov_angular_velocity[9]; IN/OUT:
 *                      without use_tf the reference ACCUMULATES into it (position +=, covariances +=, :219-222)
 *      delta [B][31] = pose record (25) + velocity[3] angular_velocity[3]; may be NULL.  delta[0:7] + [25:31] is the `u` of
 *                      SLK_PM_DELTA_POSE.
 *      use_tf != 0: tfPostPose = tfPrevPose * tfDeltaPose through slk_transform_compose's arithmetic (:202-215). ---- */
int slk_dead_reckon_pose(slk_filter *f, const double *u, int u_stride, const double *velcov, int c_stride,
                         const double *prev, double *post, double *delta, int use_tf, int where);

/* ---- AdaptiveAttitudeCov (src/filters/MeasurementModels.hpp:136-286): a batch of B independent objects (history of
 *      m1 residual outer products, r1count, r2count each) resident on the device.  slk_adaptive_matrix is one call of
 *      ::matrix(xk, Pk, z, H, R) per object: xk [B][n], Pk [B][n*n], z [B][3], H [B][3*n] (3 x n column-major),
 *      R 3x3 (r_stride 0 = shared) -> Rout [B][9], which is the R (r_stride 9) of slk_update / slk_step. ---- */
typedef struct slk_adaptive slk_adaptive;
int  slk_adaptive_create(int batch, int device, unsigned m1, unsigned m2, double gamma, unsigned r2count, void *stream,
                         slk_adaptive **out);
void slk_adaptive_destroy(slk_adaptive *a);
int  slk_adaptive_matrix(slk_adaptive *a, int n, const double *xk, const double *Pk, const double *z, const double *H,
                         const double *R, int r_stride, double *Rout, int where);

/* ---- update(z, h, R[, mt]): UKF update, Msckf.hpp:196-277 (chi-square gate per 2-row block
 *      + applyDelta re-draw) and Usckf.hpp:246-308 (whole-vector gate, direct boxplus).
 *      params [B][p_stride] model parameters (0 = shared), z [B][m], R m x m (r_stride 0 = shared).
 *      gate: Msckf 0 = accept all blocks, 1 = accept_mahalanobis_distance (Msckf.hpp:199,844-905);
 *            Usckf 0 = accept_any (Usckf.hpp:249), d = chi-square dof of the whole-vector gate. ---- */
int slk_update(slk_filter *f, int model, const double *params, int p_stride,
               const double *z, int m, const double *R, int r_stride, int gate, int where);

/* ---- update(z, h, R, mt) with an ARBITRARY significance test `mt` (Msckf.hpp:220-223, Usckf.hpp:262-302): the test is a
 *      host callable, so the update is split around it.
 *      slk_update_innovation: sigma points, Z = h(X), innovation and S = cov(Z) + R exactly as slk_update computes them,
 *        handed back as SI [B][m*m + m] = S (column-major), innovation; the filter is not modified.  `model` may be
 *        SLK_MODEL_EXTERNAL with Z [B][2N+1][m] = h(X) of slk_update_sigma_points (else Z = NULL).
 *      slk_update_selected (Msckf): the update with the surviving rows decided by the caller -- rowsel [B][m + 2] =
 *        { number of surviving rows, number of outlier blocks, surviving row indices in order } -- instead of the built-in
 *        chi-square loop; everything else as slk_update / slk_update_from_sigma.
 *      (Usckf has a whole-vector test: the caller evaluates mt on S, innovation and then calls slk_update with gate 0
 *      or not at all.) ---- */
int slk_update_innovation(slk_filter *f, int model, const double *params, int p_stride, const double *Z,
                          const double *z, int m, const double *R, int r_stride, double *SI, int where);
int slk_update_selected(slk_filter *f, int model, const double *params, int p_stride, const double *Z,
                        const double *z, int m, const double *R, int r_stride, const int *rowsel, int where);

/* ---- EKF update(z, h, H, R[, mt]): Msckf.hpp:284-349 (Msckf only).  The reference's functor h(mu_state, H) is
 *      evaluated by the caller at the current mean: zmean [B][m] = h(mu), H [B][m*N] = its Jacobian, m x N column-major
 *      per filter (Eigen default), m >= N rows (reduceDimension, :791-816, compresses to N).  R as in slk_update.
 *      gate: 0 = accept all 2-row blocks, 1 = accept_mahalanobis_distance (:285-289, :756-789 incl. its indexing of the
 *      unreduced information matrix and the shifted second erase).  Outliers: slk_get_outliers. ---- */
int slk_update_ekf(slk_filter *f, const double *z, const double *zmean, const double *H, int m,
                   const double *R, int r_stride, int gate, int where);

/* ---- fused predict + update, one kernel launch, state stays on chip between the two
 *      (the benchmark's "filter step") ---- */
int slk_step(slk_filter *f, int pmodel, const double *u, int u_stride, const double *Q, int q_stride,
             int mmodel, const double *params, int p_stride, const double *z, int m,
             const double *R, int r_stride, int gate, int where);

/* ---- Tier B (opaque host functors, the reference's boost::bind form:
 *      UsckfUnitTest.cpp:246,284; MsckfUnitTest.cpp:200-205).  The library draws the sigma
 *      points (generateSigmaPoints, Msckf.hpp:400-468 / Usckf.hpp:532-598), the caller applies
 *      f / h, the library finishes the step with the same kernels as Tier A. ---- */
int slk_predict_sigma_points(slk_filter *f, double *X /* [B][25][13] */, int where);
int slk_predict_from_sigma(slk_filter *f, const double *Y /* [B][25][13] = f(X) */,
                           const double *Q, int q_stride, int where);
int slk_update_sigma_points(slk_filter *f, double *X /* [B][2N+1][Nq] */, int where);
int slk_update_from_sigma(slk_filter *f, const double *Z /* [B][2N+1][m] = h(X) */,
                          const double *z, int m, const double *R, int r_stride, int gate, int where);

/* ---- Usckf bookkeeping: cloning() Usckf.hpp:391-433, setMeasurement() Usckf.hpp:322-389
 *      (changes N; z [B][n], R n x n shared) ---- */
int slk_usckf_cloning(slk_filter *f, int mode);
int slk_usckf_set_measurement(slk_filter *f, int mode, const double *z, int n, const double *R, int where);
/* Msckf sliding window: caller-side push/pop on muState().sensorsk + setPk (Msckf.hpp:381-395) */
int slk_msckf_resize(slk_filter *f, int n_clones);
/* ... and the same on the device, so that a trajectory never round-trips through the host (the reference has no
 *     such call: its callers push/pop muState().sensorsk and setPk, Msckf.hpp:381-395, State.hpp:342, :373-396):
 *     clone_pose appends a SensorState equal to the current pose (pos, orient) whose covariance rows / columns copy
 *     the pose's (J P J^T, J = [I; E_pose]); drop_clone removes clone `index` (0 = oldest) with its 6 rows / columns. */
int slk_msckf_clone_pose(slk_filter *f);
int slk_msckf_drop_clone(slk_filter *f, int index);

/* ---- checkSigmaPoints(): Msckf.hpp:819-839 (Usckf.hpp:769-789 is the same self test).  Re-draws the sigma points of
 *      (mu_state, Pk), takes their manifold mean and covariance on the device and reports per filter
 *      max_cov_err [B] = max |covSigmaPoints - Pk| (the reference asserts <= 1e-6) and mean_err [B] = |mu_state [-] muX|
 *      (the reference asserts mu_state == muX, i.e. isZero(1e-12) of the difference, MtkWrap.hpp:108-112).
 *      The filter is not modified.  Msckf only (the Usckf facade runs the same test on the sigma points of
 *      slk_update_sigma_points). ---- */
int slk_check_sigma_points(slk_filter *f, double *max_cov_err, double *mean_err, int where);

/* ---- arithmetic of the covariance rebuild (Msckf.hpp:665 -> :574-589): SLK_PREC_F64 (default, the
 *      parity path), SLK_PREC_F32 (fp32 MFMA) or SLK_PREC_BF16 (bf16 operands, fp32 accumulation).
 *      The reduced modes exist for the tolerance sweep of BASELINE.json config 5; the reference is
 *      double throughout (State.hpp:37-38). ---- */
enum { SLK_PREC_F64 = 0, SLK_PREC_F32 = 1, SLK_PREC_BF16 = 2 };
int slk_set_rebuild_precision(slk_filter *f, int mode);

/* ---- results of the last update / accumulated status ---- */
int slk_get_outliers(slk_filter *f, unsigned *outliers /* [B], return value of Msckf::update :276 */, int where);
int slk_get_status(slk_filter *f, int *status /* [B] */, int where);
int slk_clear_status(slk_filter *f);
int slk_sync(slk_filter *f);

/* ---- measurement aids (HIP events on the handle's stream) ---- */
int slk_timer_start(slk_filter *f);
int slk_timer_stop(slk_filter *f, float *milliseconds);

/* self test of the fp64 MFMA fragment layout used by the covariance rebuild: multiplies an
 * asymmetric 16x16 pair on the device and checks it on the host.  0 = layout as documented. */
int slk_selftest_mfma(int device);

#ifdef __cplusplus
}
#endif
#endif /* SLK_H */
