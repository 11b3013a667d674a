"""Seeded scenario / synthetic-input definitions shared by the golden generator, the CPU
oracle tests, the GPU parity tests, bench.py and smoke().  Pure numpy, no oracle, no product.

Scenario sources in the reference:
  * usckf_unit_test_*  : test/UsckfUnitTest.cpp:175-284 (USCKF_DYNAMIC)
  * msckf_unit_test_*  : test/MsckfUnitTest.cpp:151-206 (MSCKF), with the square SPD
                         Pk_0 of SURVEY.md Appendix B.2
  * synthetic_msckf    : SURVEY.md 8(d) synthetic inputs (batched Monte-Carlo filters)
"""
import numpy as np

D2R = np.pi / 180.0   # src/Configuration.hpp:36


def quat_exp(v):
    """exp map (rotation vector -> quaternion x,y,z,w), plain numpy, for input generation only."""
    v = np.asarray(v, dtype=np.float64)
    th = np.linalg.norm(v, axis=-1, keepdims=True)
    half = 0.5 * th
    k = np.where(th > 1e-12, np.sin(half) / np.where(th > 1e-12, th, 1.0), 0.5)
    return np.concatenate([k * v, np.cos(half)], axis=-1)


def quat_mul(a, b):
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def quat_rotate(q, v):
    u, w = q[..., :3], q[..., 3:4]
    uv = 2.0 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def euler_zyx_quat(z, y, x):
    """AngleAxis(z,Z)*AngleAxis(y,Y)*AngleAxis(x,X) as in MsckfUnitTest.cpp:165-168."""
    qz = quat_exp(np.array([0, 0, z]))
    qy = quat_exp(np.array([0, y, 0]))
    qx = quat_exp(np.array([x, 0, 0]))
    return quat_mul(quat_mul(qz, qy), qx)


# ------------------------------------------------------------------ reference unit-test scenarios
def usckf_unit_test():
    """Inputs of USCKF_DYNAMIC, test/UsckfUnitTest.cpp:175-284."""
    state = np.zeros(13)
    state[6] = 1.0
    return dict(
        state_single=state,
        P0_single=0.0025 * np.eye(12),                              # :186
        dt=0.01,                                                    # :182
        set_measurements=[                                          # :198-225
            (1, np.full(3, 3.34), 0.008 * np.eye(3)),               # STATEK
            (2, np.full(9, 1.34), 0.008 * np.eye(9)),               # STATEK_L
            (1, np.full(3, 3.35), 0.05 * np.eye(3)),                # STATEK again
        ],
        velocity=np.array([100.0, 0.0, 0.0]),                       # :242
        angular_velocity=np.full(3, 100.0 * D2R),                   # :243
        Q=0.1 * 0.01 * np.eye(12),                                  # :51-60 with dt
        n_predict=2,                                                # :239
        z=np.array([2.33, 3.35, 3.35]),                             # :268
        R=0.01 * np.eye(3),                                         # :280-283
    )


def msckf_unit_test(k=4):
    """Inputs of MSCKF, test/MsckfUnitTest.cpp:151-206, with a square Pk_0 = 0.025*I."""
    N = 12 + 6 * k
    mean = np.zeros(13 + 7 * k)
    mean[6] = 1.0
    for c in range(k):
        mean[13 + 7 * c + 6] = 1.0
    return dict(
        k=k, mean=mean, P=0.025 * np.eye(N),
        dpos=np.full(3, 0.1), dquat=euler_zyx_quat(D2R, D2R, D2R),  # :163-168
        velocity=np.full(3, 0.1), angular_velocity=np.full(3, 0.1),  # :169-170
        Q=0.01 * np.eye(12),                                        # :173-177
        n_predict=2,                                                # :196
    )


# ------------------------------------------------------------------ synthetic batches
def synthetic_msckf(B, k, m=8, seed=0x5EED0000, meas_sigma=0.05):
    """B independent Msckf filters with k clones and m/2 2-D features.

    Returns a dict of C-contiguous float64 arrays:
      mean [B, 13+7k], P [B, N, N] (symmetric SPD), u [B, 13] delta-pose process input
      (dpos3, dquat4, velocity3, angular_velocity3), feat [B, m/2, 4] (landmark xyz, pose index),
      z [B, m], Q [12, 12], R [m, m].
    """
    rng = np.random.default_rng(seed)
    N, Nq = 12 + 6 * k, 13 + 7 * k
    nf = m // 2
    mean = np.zeros((B, Nq))
    pos = rng.uniform(-10, 10, (B, 3))
    quat = quat_exp(rng.uniform(-np.pi / 4, np.pi / 4, (B, 3)))
    mean[:, 0:3], mean[:, 3:7] = pos, quat
    mean[:, 7:10] = rng.normal(0, 1, (B, 3))
    mean[:, 10:13] = rng.normal(0, 0.1, (B, 3))
    for c in range(k):
        s = 13 + 7 * c
        mean[:, s:s + 3] = pos + rng.normal(0, 0.05, (B, 3))
        mean[:, s + 3:s + 7] = quat_mul(quat, quat_exp(rng.normal(0, 0.05, (B, 3))))
    A = rng.normal(0, 0.05 / np.sqrt(N), (B, N, N))
    P = A @ np.transpose(A, (0, 2, 1)) + 0.01 * np.eye(N)
    P = 0.5 * (P + np.transpose(P, (0, 2, 1)))
    u = np.zeros((B, 13))
    u[:, 0:3] = rng.normal(0.1, 0.01, (B, 3))
    u[:, 3:7] = quat_exp(rng.normal(0, D2R, (B, 3)))
    u[:, 7:10] = 0.1
    u[:, 10:13] = 0.1
    feat = np.zeros((B, nf, 4))
    z = np.zeros((B, m))
    for j in range(nf):
        c = (j % k) + 1 if k > 0 else 0            # observing pose: clones round-robin (0 = statek)
        s = 0 if c == 0 else 13 + 7 * (c - 1)
        local = np.concatenate([rng.uniform(-1, 1, (B, 2)), rng.uniform(4, 8, (B, 1))], axis=1)
        feat[:, j, 0:3] = mean[:, s:s + 3] + quat_rotate(mean[:, s + 3:s + 7], local)
        feat[:, j, 3] = c
        z[:, 2 * j:2 * j + 2] = local[:, 0:2] / local[:, 2:3] + rng.normal(0, meas_sigma, (B, 2))
    return dict(B=B, k=k, m=m, N=N, Nq=Nq, mean=np.ascontiguousarray(mean), P=np.ascontiguousarray(P),
                u=np.ascontiguousarray(u), feat=np.ascontiguousarray(feat), z=np.ascontiguousarray(z),
                Q=0.01 * np.eye(12), R=0.01 * np.eye(m))


def synthetic_usckf(B, nfk=3, nfkl=9, seed=0x5EED1000):
    """B independent Usckf filters, well-posed (SPD) variant of the unit-test shape:
    N = 36 + nfk + nfkl; clones carry independent noise instead of the exactly singular
    cloned covariance of Usckf.hpp:399-426."""
    rng = np.random.default_rng(seed)
    N, Nq = 36 + nfk + nfkl, 39 + nfk + nfkl
    mean = np.zeros((B, Nq))
    pos = rng.uniform(-2, 2, (B, 3))
    quat = quat_exp(rng.uniform(-0.5, 0.5, (B, 3)))
    for s in range(3):
        o = 13 * s
        mean[:, o:o + 3] = pos + rng.normal(0, 0.1, (B, 3))
        mean[:, o + 3:o + 7] = quat_mul(quat, quat_exp(rng.normal(0, 0.05, (B, 3))))
        mean[:, o + 7:o + 10] = rng.normal(0, 0.5, (B, 3))
        mean[:, o + 10:o + 13] = rng.normal(0, 0.1, (B, 3))
    mean[:, 39:] = rng.uniform(1, 4, (B, nfk + nfkl))
    A = rng.normal(0, 0.05 / np.sqrt(N), (B, N, N))
    P = A @ np.transpose(A, (0, 2, 1)) + 0.0025 * np.eye(N)
    P = 0.5 * (P + np.transpose(P, (0, 2, 1)))
    u = np.zeros((B, 7))
    u[:, 0:3] = rng.normal(1.0, 0.1, (B, 3))
    u[:, 3:6] = rng.normal(0, 10 * D2R, (B, 3))
    u[:, 6] = 0.01
    z = mean[:, 39:39 + nfk] + rng.normal(0, 0.1, (B, nfk))
    return dict(B=B, nfk=nfk, nfkl=nfkl, N=N, Nq=Nq, mean=np.ascontiguousarray(mean), P=np.ascontiguousarray(P),
                u=np.ascontiguousarray(u), z=np.ascontiguousarray(z), Q=0.1 * 0.01 * np.eye(12),
                R=0.01 * np.eye(nfk))


def synthetic_ekf(B, k, m, seed=0xEC0F, outliers=True):
    """Inputs of the Msckf EKF update (Msckf.hpp:284-349) for B filters with k clones: what the reference's functor
    h(mu_state, H) hands back (zmean, H [m x N]) plus z and a non-isotropic R.  The Jacobians have exactly zero
    columns for velocity / angular velocity (as stacked feature residuals do) and are dense elsewhere: the exact
    zeros are handled identically by every Householder sweep (tau = 0), whereas a numerically dependent column would
    make thinQ -- and with a non-isotropic R the result -- depend on rounding noise."""
    s = synthetic_msckf(B, k, m=2, seed=seed)
    N = s["N"]
    rng = np.random.default_rng(seed + 1)
    H = rng.normal(0, 1.0, (B, m, N))
    H[:, :, 6:12] = 0.0
    A = rng.normal(0, 0.05, (B, m, m))
    R = A @ np.transpose(A, (0, 2, 1)) + 0.02 * np.eye(m)
    zmean = rng.normal(0, 1.0, (B, m))
    z = zmean + rng.normal(0, 0.15, (B, m))
    if outliers:
        for b in range(B):
            for r in rng.choice(m // 2, size=1 + b % 3, replace=False):
                z[b, 2 * r] += 25.0
    return dict(B=B, k=k, m=m, N=N, Nq=s["Nq"], mean=s["mean"], P=s["P"].reshape(B, N, N), H=H, R=R, z=z, zmean=zmean)


def synthetic_pose_ops(B=24, seed=0xD0E5):
    """Inputs of the pose-with-uncertainty ops: transform pairs with 6x6 [r t] covariances, dead
    reckoning samples with a velocity covariance, and an AdaptiveAttitudeCov measurement sequence.  Rotations stay below
    ~100 degrees, where Eigen's rotation-matrix -> quaternion conversion takes its trace > 0 branch (w > 0)."""
    rng = np.random.default_rng(seed)

    def spd(n, s, count):
        A = rng.normal(0, s, (count, n, n))
        return A @ np.transpose(A, (0, 2, 1)) + 1e-4 * np.eye(n)
    t2 = np.concatenate([rng.normal(0, 2, (B, 3)), quat_exp(rng.normal(0, 0.5, (B, 3)))], axis=1)
    t1 = np.concatenate([rng.normal(0, 1, (B, 3)), quat_exp(rng.normal(0, 0.4, (B, 3)))], axis=1)
    u = np.concatenate([0.01 + 0.1 * rng.random((B, 1)), rng.normal(0, 1, (B, 3)), rng.normal(0, 0.5, (B, 3)),
                        rng.normal(0, 1, (B, 3)), rng.normal(0, 0.5, (B, 3))], axis=1)
    prev = np.concatenate([rng.normal(0, 3, (B, 3)), quat_exp(rng.normal(0, 0.5, (B, 3))),
                           np.transpose(spd(3, 0.1, B), (0, 2, 1)).reshape(B, 9),
                           np.transpose(spd(3, 0.05, B), (0, 2, 1)).reshape(B, 9)], axis=1)
    n, steps = 12, 30
    H = np.zeros((steps, B, 3, n))
    H[:, :, :, 3:6] = np.eye(3)
    H += rng.normal(0, 0.01, H.shape)
    xk = rng.normal(0, 0.1, (steps, B, n))
    noise = np.where((np.arange(steps) % 7 == 0)[:, None, None], 0.3, 0.03)
    z = np.einsum("sbij,sbj->sbi", H, xk) + rng.normal(0, 1, (steps, B, 3)) * noise
    return dict(B=B, t2=t2, t1=t1, cov2=spd(6, 0.05, B), cov1=spd(6, 0.03, B), u=u, velcov=spd(6, 0.1, 1)[0], prev=prev,
                n=n, steps=steps, xk=xk, Pk=spd(n, 0.02, B), z=z, H=H, R=spd(3, 0.02, 1)[0],
                m1=5, m2=3, gamma=0.002, r2count=0)
