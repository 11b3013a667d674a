"""CPU: oracle restatements of the pose-with-uncertainty ops next to the filter hot path against
the committed golden vectors, the independent numpy twin and closed-form known answers.
  TransformWithUncertainty::operator*      src/core/Transform.cpp:215-254 (Jacobians :35-137)
  DeadReckon::updatePose (both overloads)  src/core/DeadReckon.hpp:129-239, :306-330
  AdaptiveAttitudeCov::matrix              src/filters/MeasurementModels.hpp:181-286
"parity unpinned" with respect to the reference itself: it holds no vectors for these functions either."""
import os

import numpy as np

from oracle import np_check as npc
from oracle import oracle as o
import scenarios as sc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def same_transform(a, b):
    return max(np.abs(a[:3] - b[:3]).max(), min(np.abs(a[3:] - b[3:]).max(), np.abs(a[3:] + b[3:]).max()))


def test_transform_compose_golden():
    g = np.load(os.path.join(G, "pose_ops.npz"))
    s = sc.synthetic_pose_ops()
    for name, use2, use1 in (("both", True, True), ("first", True, False), ("second", False, True), ("none", False, False)):
        for b in range(s["B"]):
            t, c = o.transform_compose(s["t2"][b], s["cov2"][b] if use2 else None, s["t1"][b], s["cov1"][b] if use1 else None)
            assert np.abs(t - g[f"compose_{name}_t"][b]).max() <= 1e-15
            assert np.abs(c - g[f"compose_{name}_cov"][b]).max() <= 1e-15
    assert np.abs(g["compose_none_cov"]).max() == 0.0          # no uncertainty in, none out (Transform.cpp:219-220)


def test_transform_compose_closed_form():
    # identity rotations: the composition is a translation sum; a rotational uncertainty of the left transform moves the
    # translated point by -[x]x (drx_by_dr at the identity), everything else adds up
    rng = np.random.default_rng(1)
    x = np.array([0.7, -1.2, 2.0])
    t2 = np.r_[1.0, 2.0, 3.0, 0, 0, 0, 1.0]
    t1 = np.r_[x, 0, 0, 0, 1.0]
    A, Bm = rng.normal(0, 0.1, (6, 6)), rng.normal(0, 0.1, (6, 6))
    c2, c1 = A @ A.T, Bm @ Bm.T
    t, c = o.transform_compose(t2, c2, t1, c1)
    S = np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]])
    J2 = np.eye(6)
    J2[3:, :3] = -S
    assert np.abs(t - np.r_[1.7, 0.8, 5.0, 0, 0, 0, 1.0]).max() <= 1e-15
    assert np.abs(c - (c1 + J2 @ c2 @ J2.T)).max() <= 1e-15
    # composition of the poses themselves is associative and agrees with plain quaternion algebra at any angle,
    # including the branch of Eigen's matrix -> quaternion conversion with a non-positive trace
    for ang in (0.3, 2.0, 3.0):
        a = np.r_[rng.normal(size=3), o.so3_exp(np.array([ang, 0.2, -0.1]))]
        b = np.r_[rng.normal(size=3), o.so3_exp(np.array([0.1, ang, 0.3]))]
        t, _ = o.transform_compose(a, None, b, None)
        q = o.quat_mul(a[3:], b[3:])
        assert same_transform(t, np.r_[a[:3] + o.quat_rotate(a[3:], b[:3]), q]) <= 1e-14


def test_dead_reckon_pose_golden_and_branches():
    g = np.load(os.path.join(G, "pose_ops.npz"))
    s = sc.synthetic_pose_ops()
    for tf in (0, 1):
        for b in range(s["B"]):
            post = np.r_[s["prev"][b], np.zeros(24)]
            po, de = o.dead_reckon_pose(s["u"][b], s["velcov"], s["prev"][b], post, tf)
            assert np.abs(po - g[f"dr_pose_tf{tf}_post"][b]).max() <= 1e-15
            assert np.abs(de - g[f"dr_pose_tf{tf}_delta"][b]).max() <= 1e-15
            # the delta pose is the dead-reckoning delta of slk_dead_reckon (round 1), the covariances are C dt^2
            d13 = o.dead_reckon_delta(s["u"][b])
            assert np.array_equal(de[:7], np.ravel(d13)[:7]) and np.array_equal(de[25:], np.ravel(d13)[7:])
            dt = s["u"][b, 0]
            assert np.abs(de[7:16].reshape(3, 3).T - s["velcov"][:3, :3] * dt * dt).max() <= 1e-18
            assert np.abs(de[16:25].reshape(3, 3).T - s["velcov"][3:, 3:] * dt * dt).max() <= 1e-18
    # without TransformWithUncertainty the call ACCUMULATES into postPose (DeadReckon.hpp:219-222)
    b = 3
    post = np.r_[s["prev"][b], np.zeros(24)]
    p1, d1 = o.dead_reckon_pose(s["u"][b], s["velcov"], s["prev"][b], post, 0)
    p2, _ = o.dead_reckon_pose(s["u"][b], s["velcov"], s["prev"][b], p1, 0)
    assert np.abs((p2[:3] - p1[:3]) - (p1[:3] - s["prev"][b, :3])).max() <= 1e-14
    assert np.abs((p2[7:16] - p1[7:16]) - d1[7:16]).max() <= 1e-16
    # a NaN anywhere in the velocity covariance zeroes the delta covariances (:165-176)
    vc = s["velcov"].copy()
    vc[4, 1] = np.nan
    _, dn = o.dead_reckon_pose(s["u"][b], vc, s["prev"][b], post, 0)
    assert np.abs(dn[7:25]).max() == 0.0
    # the Affine3d overload (:306-330)
    pc, dc = s["cov2"][b], s["cov1"][b]
    t, c = o.update_pose_affine(s["t2"][b], pc, s["t1"][b], dc, 0)
    t0, _ = o.transform_compose(s["t2"][b], None, s["t1"][b], None)
    assert np.array_equal(t, t0) and np.abs(c - (pc + dc)).max() <= 1e-18
    t, c = o.update_pose_affine(s["t2"][b], pc, s["t1"][b], dc, 1)
    t1, c1 = o.transform_compose(s["t2"][b], pc, s["t1"][b], dc)
    assert np.array_equal(t, t1) and np.array_equal(c, c1)


def test_adaptive_attitude_cov_golden_and_properties():
    g = np.load(os.path.join(G, "pose_ops.npz"))
    s = sc.synthetic_pose_ops()
    B = s["B"]
    objs = [o.AdaptiveAttitudeCov(s["m1"], s["m2"], s["gamma"], s["r2count"]) for _ in range(B)]
    grew = 0
    for k in range(s["steps"]):
        for b in range(B):
            R = objs[b].matrix(s["xk"][k, b], s["Pk"][b], s["z"][k, b], s["H"][k, b], s["R"])
            assert np.abs(R - g["adaptive_R"][k, b]).max() <= 1e-15
            Q = R - s["R"]
            assert np.abs(Q - Q.T).max() <= 1e-15 and np.linalg.eigvalsh(Q).min() >= -1e-15     # R + a PSD matrix
            grew += Q.max() > 1e-6
    assert grew > 0 and grew < s["steps"] * B               # both branches are exercised
    # closed form: m1 = 1, a residual far above H P H^T + R -> Qstar = (|r|^2 - u^T fooR u) u u^T with u = r / |r|
    a = o.AdaptiveAttitudeCov(1, 2, 0.01, 0)
    n = 3
    r = np.array([0.9, -0.4, 0.2])
    P, Rm = 0.01 * np.eye(n), 0.02 * np.eye(3)
    Rn = a.matrix(np.zeros(n), P, r, np.eye(3), Rm)
    uu = r / np.linalg.norm(r)
    want = Rm + (r @ r - 0.03) * np.outer(uu, uu)
    assert np.abs(Rn - want).max() <= 1e-14
    # below gamma: r2count counts up and after m2 quiet calls Qstar is dropped
    a = o.AdaptiveAttitudeCov(1, 2, 10.0, 0)
    assert np.abs(a.matrix(np.zeros(n), P, r, np.eye(3), Rm) - want).max() <= 1e-14     # r2count 1 < m2
    assert np.abs(a.matrix(np.zeros(n), P, r, np.eye(3), Rm) - Rm).max() == 0.0         # r2count 2: plain R
