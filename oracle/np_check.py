"""Independent numpy/scipy re-implementation of the sigma-point filters (G6 of SURVEY.md 8c).

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/slk_oracle.c, on top of
scipy.spatial.transform.Rotation / numpy.linalg, so that an error in the C oracle's
SO(3), Cholesky, inverse or bookkeeping does not silently pass: tests require the two
to agree to <= 1e-12 relative on every golden scenario.  It follows the same reference
sources (src/filters/Msckf.hpp, Usckf.hpp, State.hpp) but shares no code with the oracle.
"""
import numpy as np
from scipy.spatial.transform import Rotation as Rot

CHI2_95 = {1: 3.84, 2: 5.99, 3: 7.81, 4: 9.49, 5: 11.07, 6: 12.59, 7: 14.07, 8: 15.51, 9: 16.92}


class Manifold:
    """A product of vector blocks and SO(3) blocks; storage quaternion order (x,y,z,w)."""

    def __init__(self, blocks):
        # blocks: list of (is_so3, tangent_offset, storage_offset, length)
        self.blocks = blocks
        self.N = max(t + l for _, t, _, l in blocks)
        self.Nq = max(s + (4 if so3 else l) for so3, _, s, l in blocks)

    @staticmethod
    def _state_blocks(t0, s0):
        return [(False, t0, s0, 3), (True, t0 + 3, s0 + 3, 3), (False, t0 + 6, s0 + 7, 3), (False, t0 + 9, s0 + 10, 3)]

    @classmethod
    def single(cls):
        return cls(cls._state_blocks(0, 0))

    @classmethod
    def multi(cls, k):
        b = cls._state_blocks(0, 0)
        for c in range(k):
            b += [(False, 12 + 6 * c, 13 + 7 * c, 3), (True, 15 + 6 * c, 16 + 7 * c, 3)]
        return cls(b)

    @classmethod
    def augmented(cls, nfk, nfkl):
        b = []
        for s in range(3):
            b += cls._state_blocks(12 * s, 13 * s)
        if nfk:
            b.append((False, 36, 39, nfk))
        if nfkl:
            b.append((False, 36 + nfk, 39 + nfk, nfkl))
        m = cls(b)
        m.N, m.Nq = 36 + nfk + nfkl, 39 + nfk + nfkl
        return m

    def plus(self, x, v):
        out = np.array(x, dtype=float)
        for so3, t, s, l in self.blocks:
            if so3:
                out[s:s + 4] = (Rot.from_quat(x[s:s + 4]) * Rot.from_rotvec(v[t:t + 3])).as_quat()
                # scipy canonicalises nothing here; keep the sign continuous with the product
            else:
                out[s:s + l] = x[s:s + l] + v[t:t + l]
        return out

    def minus(self, a, b):
        out = np.zeros(self.N)
        for so3, t, s, l in self.blocks:
            if so3:
                out[t:t + 3] = (Rot.from_quat(b[s:s + 4]).inv() * Rot.from_quat(a[s:s + 4])).as_rotvec()
            else:
                out[t:t + l] = a[s:s + l] - b[s:s + l]
        return out

    def mean(self, X):
        ref = X[0].copy()
        it = 0
        while True:
            md = np.mean([self.minus(x, ref) for x in X], axis=0)
            ref = self.plus(ref, md)
            it += 1
            if not (np.linalg.norm(md) > 1e-6 and it < 10000):
                break
        return ref, it

    def sigma(self, mu, delta, P):
        L = np.linalg.cholesky(P)
        X = [self.plus(mu, delta)]
        for j in range(self.N):
            X.append(self.plus(mu, delta + L[:, j]))
            X.append(self.plus(mu, delta - L[:, j]))
        return X

    def cov(self, mean, X):
        D = np.array([self.minus(x, mean) for x in X])
        return 0.5 * D.T @ D


# ---------------------------------------------------------------- models
def pm_const_velocity(x, velocity, angular_velocity, dt):
    y = np.zeros(13)
    y[3:7] = (Rot.from_quat(x[3:7]) * Rot.from_rotvec(np.asarray(angular_velocity) * dt)).as_quat()
    y[10:13] = angular_velocity
    y[7:10] = velocity
    y[0:3] = x[0:3] + x[7:10] * dt
    return y


def pm_delta_pose(x, dpos, dquat, velocity, angular_velocity):
    y = np.zeros(13)
    r = Rot.from_quat(x[3:7]) * Rot.from_quat(dquat)
    y[3:7] = r.as_quat()
    y[10:13] = angular_velocity
    y[0:3] = x[0:3] + r.apply(dpos)
    y[7:10] = velocity
    return y


def _omega4(w):
    """4x4 angular-velocity matrix acting on (w, x, y, z), as written in src/core/DeadReckon.hpp:259-267."""
    return np.array([[0.0, -w[0], -w[1], -w[2]],
                     [w[0], 0.0, w[2], -w[1]],
                     [w[1], -w[2], 0.0, w[0]],
                     [w[2], w[1], -w[0], 0.0]])


def update_attitude(dt, w0, w1):
    """DeadReckon::updateAttitude (src/core/DeadReckon.hpp:246-286) with the full 4x4 expression; returns (x,y,z,w)."""
    w0, w1 = np.asarray(w0, float), np.asarray(w1, float)
    om, old, eye = _omega4(w0), _omega4(w1), np.eye(4)
    n2 = float(w0 @ w0)
    M = (eye + 0.75 * om * dt - 0.25 * old * dt - (1.0 / 6.0) * n2 * dt ** 2 * eye
         - (1.0 / 24.0) * om @ old * dt ** 2 - (1.0 / 48.0) * n2 * om * dt ** 3)
    q = M @ np.array([1.0, 0.0, 0.0, 0.0])
    q = q / np.linalg.norm(q)
    return np.array([q[1], q[2], q[3], q[0]])


def dead_reckon_delta(u):
    """DeadReckon::updatePose delta pose (src/core/DeadReckon.hpp:129-239): u = dt v0 w0 v1 w1 -> dpos dquat v w."""
    u = np.asarray(u, float)
    dt, v0, w0, v1, w1 = u[0], u[1:4], u[4:7], u[7:10], u[10:13]
    return np.concatenate([(dt / 2.0) * (v0 + v1), update_attitude(dt, w0, w1), v0, w0])


def pm_dead_reckon(x, u):
    d = dead_reckon_delta(u)
    return pm_delta_pose(x, d[0:3], d[3:7], d[7:10], d[10:13])


def mm_vo_relative(X, nfk):
    single = Manifold.single()
    d = single.minus(X[0:13], X[26:39])
    R = Rot.from_rotvec(d[3:6])
    z = X[39:39 + nfk].copy()
    for i in range(0, nfk - 2, 3):
        z[i:i + 3] = R.apply(X[39 + i:39 + i + 3]) + d[0:3]
    return z


def _pose(X, c, kind):
    if kind == "multi":
        s = 0 if c == 0 else 13 + 7 * (c - 1)
    else:
        s = 13 * c
    return X[s:s + 3], X[s + 3:s + 7]


def mm_feature_proj(X, feat, kind="multi"):
    feat = np.asarray(feat, dtype=float).reshape(-1, 4)
    z = []
    for lx, ly, lz, c in feat:
        p, q = _pose(X, int(c), kind)
        loc = Rot.from_quat(q).inv().apply(np.array([lx, ly, lz]) - p)
        z += [loc[0] / loc[2], loc[1] / loc[2]]
    return np.array(z)


def mm_pose_position(X, c, kind="multi"):
    return _pose(X, int(c), kind)[0].copy()


# ---------------------------------------------------------------- filters
def predict_single(state, Pi, f, Q):
    man = Manifold.single()
    X0 = man.sigma(state, np.zeros(12), Pi)
    X1 = [f(x) for x in X0]
    mean_new, it = man.mean(X1)
    Dx = np.array([man.minus(x, state) for x in X0])
    Dy = np.array([man.minus(x, mean_new) for x in X1])
    Pxy = 0.5 * Dx.T @ Dy
    Fk = Pxy.T @ np.linalg.inv(Pi)
    return mean_new, 0.5 * Dy.T @ Dy + Q, Fk, it


class Msckf:
    def __init__(self, k, mean, P):
        self.man = Manifold.multi(k)
        self.mean = np.array(mean, dtype=float)
        self.P = np.array(P, dtype=float)

    def predict(self, f, Q):
        s, Pi, self.Fk, it = predict_single(self.mean[:13], self.P[:12, :12], f, Q)
        self.mean[:13] = s
        self.P[:12, :12] = Pi
        return it

    def update(self, z, h, R, gate=True):
        man, N = self.man, self.man.N
        X = man.sigma(self.mean, np.zeros(N), self.P)
        Z = np.array([h(x) for x in X])
        zbar = Z.sum(axis=0) / len(Z)
        innov = np.asarray(z, dtype=float) - zbar
        dZ = Z - zbar
        S = 0.5 * dZ.T @ dZ + R
        D = np.array([man.minus(x, self.mean) for x in X])
        C = 0.5 * D.T @ dZ
        # removeOutliers with the reference's shifted second erase (Msckf.hpp:741-744)
        idx = list(range(len(innov)))
        outliers, i = 0, 0

        def erase(lst, pos):
            if pos < len(lst) - 1:
                del lst[pos]
            else:
                del lst[-1]

        while i < len(idx) // 2:
            a, b = idx[2 * i], idx[2 * i + 1]
            r = innov[[a, b]]
            d2 = r @ np.linalg.inv(S[np.ix_([a, b], [a, b])]) @ r
            if gate and not d2 < CHI2_95[2]:
                erase(idx, 2 * i)
                erase(idx, 2 * i + 1)
                outliers += 1
            else:
                i += 1
        if idx:
            Sr = S[np.ix_(idx, idx)]
            K = C[:, idx] @ np.linalg.inv(Sr)
            self.P = self.P - K @ Sr @ K.T
            delta = K @ innov[idx]
            X = man.sigma(self.mean, delta, self.P)
            self.mean, it = man.mean(X)
            self.P = man.cov(self.mean, X)
        return outliers


def msckf_update_ekf(filt, z, zmean, H, R, gate=True):
    """Msckf EKF update (Msckf.hpp:284-349) with numpy/LAPACK: removeOutliers :756-789 (the information matrix is
    inverted once and indexed with the running block number), reduceDimension :791-816 through numpy's Householder
    QR (LAPACK dgeqrf uses the same reflector convention as Eigen's makeHouseholder)."""
    man, N = filt.man, filt.man.N
    H = np.array(H, dtype=float)
    R = np.array(R, dtype=float)
    innov = np.asarray(z, dtype=float) - np.asarray(zmean, dtype=float)
    info = np.linalg.inv(H @ filt.P @ H.T + R)
    idx = list(range(len(innov)))
    outliers, i = 0, 0

    def erase(lst, pos):
        if pos < len(lst) - 1:
            del lst[pos]
        else:
            del lst[-1]

    while i < len(idx) // 2:
        r = innov[[idx[2 * i], idx[2 * i + 1]]]
        d2 = r @ info[2 * i:2 * i + 2, 2 * i:2 * i + 2] @ r
        if gate and not d2 < CHI2_95[2]:
            erase(idx, 2 * i)
            erase(idx, 2 * i + 1)
            outliers += 1
        else:
            i += 1
    if idx and len(idx) < N:
        return outliers, "rows"
    if idx:
        Hq, Rq, rq = H[idx], R[np.ix_(idx, idx)], innov[idx]
        Qf, Rf = np.linalg.qr(Hq, mode="complete")
        thinQ = Qf[:, :N]
        Hr = np.triu(Rf)[:N, :N]
        rn = thinQ.T @ rq
        Rn = thinQ.T @ Rq @ thinQ
        S = Hr @ filt.P @ Hr.T + Rn
        K = filt.P @ Hr.T @ np.linalg.inv(S)
        filt.P = filt.P - K @ S @ K.T
        filt.mean = man.plus(filt.mean, K @ rn)
    return outliers, None


class Usckf:
    def __init__(self, nfk, nfkl, mean, P):
        self.nfk, self.nfkl = nfk, nfkl
        self.mean = np.array(mean, dtype=float)
        self.P = np.array(P, dtype=float)

    @property
    def man(self):
        return Manifold.augmented(self.nfk, self.nfkl)

    def predict(self, f, Q):
        s, Pi, Fk, it = predict_single(self.mean[26:39], self.P[24:36, 24:36], f, Q)
        P = self.P
        self.mean[26:39] = s
        P[24:36, 24:36] = Pi
        P[0:12, 24:36] = P[0:12, 24:36] @ Fk.T
        P[12:24, 24:36] = P[12:24, 24:36] @ Fk.T
        P[24:36, 0:12] = Fk @ P[24:36, 0:12]
        P[24:36, 12:24] = Fk @ P[24:36, 12:24]
        if self.nfk + self.nfkl:
            P[24:36, 36:] = Fk @ P[24:36, 36:]
            P[36:, 24:36] = P[24:36, 36:].T
        return it

    def update(self, z, h, R):
        man, N = self.man, self.man.N
        X = man.sigma(self.mean, np.zeros(N), self.P)
        Z = np.array([h(x) for x in X])
        zbar = Z.sum(axis=0) / len(Z)
        dZ = Z - zbar
        S = 0.5 * dZ.T @ dZ + R
        D = np.array([man.minus(x, self.mean) for x in X])
        K = (0.5 * D.T @ dZ) @ np.linalg.inv(S)
        self.P = self.P - K @ S @ K.T
        self.mean = man.plus(self.mean, K @ (np.asarray(z, dtype=float) - zbar))


# ---------------------------------------------------------------- TransformWithUncertainty / DeadReckon pose legs
# Independent of slk_oracle.c: scipy Rotation for every rotation conversion, numpy block algebra for the Jacobians of
# Pennec & Thirion as src/core/Transform.cpp:35-137 writes them.  Quaternions inside the Jacobians are (w, x, y, z).
def _canon(q_xyzw):
    """the quaternion Eigen derives from a rotation matrix on the trace > 0 branch (w > 0)"""
    q = Rot.from_quat(q_xyzw).as_quat()
    return q if q[3] >= 0 else -q


def _rvec(q_xyzw):
    """q_to_r (Transform.cpp:44-48): axis * angle with the angle in [0, pi]"""
    return Rot.from_quat(q_xyzw).as_rotvec()


def _skew(r):
    return np.array([[0.0, -r[2], r[1]], [r[2], 0.0, -r[0]], [-r[1], r[0], 0.0]])


def _dq_by_dr(q):
    r = _rvec(q)
    th2 = float(r @ r)
    return np.vstack([-q[:3] / 2.0, (0.5 - th2 / 48.0) * np.eye(3) - (1.0 / 24.0) * (1.0 - th2 / 40.0) * np.outer(r, r)])


def _dr_by_dq(q):
    v = q[:3]
    mu2 = float(v @ v)
    sg = 1.0 if q[3] > 0 else -1.0
    tau, nu = 2.0 * sg * (1.0 + mu2 / 6.0), -2.0 * sg * (2.0 / 3.0 + mu2 / 5.0)
    return np.hstack([(-2.0 * v)[:, None], tau * np.eye(3) + nu * np.outer(v, v)])


def _dq2q1(q, sgn):
    v = q[:3]
    M = np.zeros((4, 4))
    M[0, 1:], M[1:, 0], M[1:, 1:] = -v, v, sgn * _skew(v)
    return q[3] * np.eye(4) + M


def _drx_by_dr(q, x):
    r = _rvec(q)
    th2 = float(r @ r)
    al, be, ga, de = 1.0 - th2 / 6.0, 0.5 - th2 / 24.0, 1.0 / 3.0 - th2 / 30.0, -1.0 / 12.0 + th2 / 180.0
    rr, Sx, Sr = np.outer(r, r), _skew(x), _skew(r)
    return -Sx @ (ga * rr - be * Sr + al * np.eye(3)) - Sr @ Sx @ (de * rr + 2.0 * be * np.eye(3))


def transform_compose(t2, cov2, t1, cov1):
    """TransformWithUncertainty::operator* (Transform.cpp:215-254): t = pos[3] quat[4], cov 6x6 [r t] or None."""
    t2, t1 = np.asarray(t2, float), np.asarray(t1, float)
    R2, R1 = Rot.from_quat(t2[3:7]), Rot.from_quat(t1[3:7])
    out = np.concatenate([R2.apply(t1[0:3]) + t2[0:3], _canon((R2 * R1).as_quat())])
    cov = np.zeros((6, 6))
    if cov1 is None and cov2 is None:
        return out, cov
    q1, q2 = _canon(t1[3:7]), _canon(t2[3:7])
    q = (Rot.from_quat(q2) * Rot.from_quat(q1)).as_quat()
    if np.dot(q, np.r_[q2[3] * q1[:3] + q1[3] * q2[:3] + np.cross(q2[:3], q1[:3]), q2[3] * q1[3] - q2[:3] @ q1[:3]]) < 0:
        q = -q                                                   # the plain product q2 * q1 (no canonicalisation)
    if cov1 is not None:
        J1 = np.zeros((6, 6))
        J1[:3, :3] = _dr_by_dq(q) @ _dq2q1(q2, 1.0) @ _dq_by_dr(q1)
        J1[3:, 3:] = R2.as_matrix()
        cov += J1 @ np.asarray(cov1, float) @ J1.T
    if cov2 is not None:
        J2 = np.eye(6)
        J2[:3, :3] = _dr_by_dq(q) @ _dq2q1(q1, -1.0) @ _dq_by_dr(q2)
        J2[3:, :3] = _drx_by_dr(q2, t1[0:3])
        cov += J2 @ np.asarray(cov2, float) @ J2.T
    return out, cov


def dead_reckon_pose(u, velcov, prev, post, use_tf):
    """DeadReckon::updatePose, RigidBodyState overload (DeadReckon.hpp:129-239); records as in slk_oracle.c."""
    u, velcov, prev, post = (np.asarray(a, float) for a in (u, velcov, prev, post))
    dt = u[0]
    d13 = dead_reckon_delta(u)
    nan = np.isnan(velcov).any()
    cp = np.zeros((3, 3)) if nan else velcov[:3, :3] * dt * dt
    co = np.zeros((3, 3)) if nan else velcov[3:, 3:] * dt * dt
    delta = np.concatenate([d13[0:7], cp.T.ravel(), co.T.ravel(), d13[7:10], d13[10:13]])
    out = post.copy()
    pcp, pco = prev[7:16].reshape(3, 3).T, prev[16:25].reshape(3, 3).T
    if use_tf:
        z = np.zeros((3, 3))
        t, c = transform_compose(prev[0:7], np.block([[pco, z], [z, pcp]]), d13[0:7], np.block([[co, z], [z, cp]]))
        out[0:7] = t
        out[16:25], out[7:16] = c[:3, :3].T.ravel(), c[3:, 3:].T.ravel()
    else:
        out[0:3] = post[0:3] + Rot.from_quat(prev[3:7]).apply(d13[0:3])
        out[7:16] = post[7:16] + cp.T.ravel()
        out[16:25] = post[16:25] + co.T.ravel()
        qa, qb = prev[3:7], d13[3:7]
        out[3:7] = np.r_[qa[3] * qb[:3] + qb[3] * qa[:3] + np.cross(qa[:3], qb[:3]), qa[3] * qb[3] - qa[:3] @ qb[:3]]
    out[25:28], out[37:40] = u[1:4], u[4:7]
    out[28:37], out[40:49] = velcov[:3, :3].T.ravel(), velcov[3:, 3:].T.ravel()
    return out, delta


class AdaptiveAttitudeCov:
    """AdaptiveAttitudeCov (src/filters/MeasurementModels.hpp:136-286) with numpy.linalg.svd in the place of JacobiSVD."""

    def __init__(self, m1, m2, gamma, r2count):
        self.m1, self.m2, self.gamma, self.r1count, self.r2count = m1, m2, gamma, 0, r2count
        self.hist = np.zeros((m1, 3, 3))

    def matrix(self, xk, Pk, z, H, R):
        res = z - H @ xk
        self.hist[self.r1count] = np.outer(res, res)
        self.r1count = (self.r1count + 1) % self.m1
        Uk = self.hist.sum(axis=0) / self.m1
        fooR = H @ Pk @ H.T + R
        u, s, _ = np.linalg.svd(Uk)
        mu = np.array([u[:, c] @ fooR @ u[:, c] for c in range(3)])
        w = np.maximum(s - mu, 0.0)
        if (s - mu).max() > self.gamma:
            self.r2count = 0
            use = True
        else:
            self.r2count += 1
            use = self.r2count < self.m2
        Q = sum(w[c] * np.outer(u[:, c], u[:, c]) for c in range(3)) if use else np.zeros((3, 3))
        return R + Q
