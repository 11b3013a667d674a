"""Reference-independent known-answer tests for the CPU oracle and the
reference quirks it reproduces (Appendix B)."""
import numpy as np
import pytest

from oracle import oracle as o
import scenarios as sc


def spd(n, rng, scale=0.05, floor=0.01):
    A = rng.normal(0, scale, (n, n))
    return A @ A.T + floor * np.eye(n)


@pytest.mark.parametrize("k", [0, 2, 8])
def test_check_sigma_points_invariant(k):
    # G2 = Msckf::checkSigmaPoints (Msckf.hpp:819-839): cov(sigma(mu, P)) == P and mean == mu
    rng = np.random.default_rng(10 + k)
    lay = o.layout(o.MULTI, k)
    N = o.dof(lay)
    mu = o.set_from_vector(lay, rng.normal(size=N))
    P = spd(N, rng)
    st, cov_err, mean_err = o.Msckf(k, mu, P).check_sigma_points()
    assert st == 0
    assert cov_err <= 1e-12 * np.abs(P).max() + 1e-15      # reference tolerance is 1e-6 absolute
    assert mean_err <= 1e-12


def test_cholesky_and_inverse():
    rng = np.random.default_rng(3)
    A = spd(20, rng)
    L, fail = o.cholesky_lower(A)
    assert fail == -1
    np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-12, atol=1e-15)
    Ai, sing = o.inverse(A)
    assert sing == 0
    np.testing.assert_allclose(Ai @ A, np.eye(20), atol=1e-10)
    # failing pivot is reported (the reference ignores it: Usckf.hpp:537-538)
    B = np.array([[1.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    _, fail = o.cholesky_lower(B)
    assert fail == 1


def test_linear_kat_predict():
    # G3: const-velocity model with omega = 0 is affine in the tangent space:
    # pos' = pos + velo*dt, orient' = orient, velo' = const, angvelo' = const
    # => P' = F P F^T + Q exactly (1/2 sum (+-L_j)(+-L_j)^T = L L^T).
    rng = np.random.default_rng(4)
    dt = 0.1
    single = o.layout(o.SINGLE)
    mu = o.set_from_vector(single, rng.normal(size=12))
    P = spd(12, rng)
    Q = 0.01 * np.eye(12)
    f = o.Msckf(0, mu, P)
    assert f.predict(o.pm_const_velocity([0.3, -0.2, 0.1], [0, 0, 0], dt), Q) == 0
    F = np.zeros((12, 12))
    F[0:3, 0:3] = np.eye(3)
    F[0:3, 6:9] = dt * np.eye(3)
    F[3:6, 3:6] = np.eye(3)
    np.testing.assert_allclose(f.P, F @ P @ F.T + Q, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(f.Fk, F, atol=1e-12)          # Fk = Pxy^T P^-1 recovers the Jacobian
    expect = mu.copy()
    expect[0:3] += mu[7:10] * dt
    expect[7:10] = [0.3, -0.2, 0.1]
    expect[10:13] = 0
    np.testing.assert_allclose(o.boxminus(single, f.mean, expect), 0, atol=1e-13)


def test_linear_kat_update():
    # G3: z = position of clone 1, P block-diagonal with rotations decoupled => closed-form
    # Kalman update; applyDelta's re-drawn sigma points reproduce (mu + K nu, P - K S K^T).
    rng = np.random.default_rng(5)
    k = 2
    lay = o.layout(o.MULTI, k)
    N = o.dof(lay)
    mu = o.set_from_vector(lay, rng.normal(size=N))
    vec_idx = [i for i in range(N) if not (3 <= i < 6 or (i >= 12 and (i - 12) % 6 >= 3))]
    rot_idx = [i for i in range(N) if i not in vec_idx]
    P = np.zeros((N, N))
    P[np.ix_(vec_idx, vec_idx)] = spd(len(vec_idx), rng)
    P[np.ix_(rot_idx, rot_idx)] = spd(len(rot_idx), rng, scale=0.02)
    R = 0.02 * np.eye(3)
    H = np.zeros((3, N))
    H[:, 12:15] = np.eye(3)
    z = mu[13:16] + np.array([0.05, -0.02, 0.01])
    f = o.Msckf(k, mu, P)
    st, no = f.update(z, o.mm_pose_position(1), R, gate=False)
    assert st == 0 and no == 0
    S = H @ P @ H.T + R
    K = P @ H.T @ np.linalg.inv(S)
    np.testing.assert_allclose(f.P, P - K @ S @ K.T, rtol=1e-11, atol=1e-14)
    expect = o.boxplus(lay, mu, K @ (z - mu[13:16]))
    np.testing.assert_allclose(o.boxminus(lay, f.mean, expect), 0, atol=1e-12)


def test_chi2_gate_table():
    # Msckf.hpp:844-905
    thr = {1: 3.84, 2: 5.99, 3: 7.81, 4: 9.49, 5: 11.07, 6: 12.59, 7: 14.07, 8: 15.51, 9: 16.92}
    L = o.lib()
    for dof, t in thr.items():
        assert L.slko_accept_mahalanobis(t - 1e-9, dof) == 1
        assert L.slko_accept_mahalanobis(t, dof) == 0
    assert L.slko_accept_mahalanobis(0.0, 10) == 0 and L.slko_accept_mahalanobis(0.0, 0) == 0


def test_remove_outliers_shifted_second_erase():
    # Appendix B.4 / Msckf.hpp:741-744: rejecting 2-D feature i erases ORIGINAL rows 2i and 2i+2
    # (not 2i+1).  Use the position model per clone so the surviving rows are identifiable:
    # measurement rows = [p1.x p1.y | p2.x p2.y | p3.x p3.y]; reject block 0 => rows {1,3,4,5} survive.
    rng = np.random.default_rng(6)
    k = 3
    lay = o.layout(o.MULTI, k)
    N = o.dof(lay)
    mu = o.identity_state(lay)
    P = 0.01 * np.eye(N)
    R = 0.01 * np.eye(6)

    def h(X):
        return np.array([X[13], X[14], X[20], X[21], X[27], X[28]])

    z = np.zeros(6)
    z[0] = 5.0                                              # gross outlier in block 0 only
    z[1], z[3], z[4], z[5] = 0.05, -0.04, 0.03, 0.02
    z[2] = 0.07                                             # original row 2 is (wrongly) erased with block 0
    f = o.Msckf(k, mu, P)
    st, no = f.update(z, o.mm_python(h), R, gate=True)
    assert st == 0 and no == 1
    # closed form with surviving rows {1,3,4,5}
    rows = [1, 3, 4, 5]
    H = np.zeros((6, N))
    for r, c in enumerate([12, 13, 18, 19, 24, 25]):
        H[r, c] = 1.0
    Hs = H[rows]
    S = Hs @ P @ Hs.T + R[np.ix_(rows, rows)]
    K = P @ Hs.T @ np.linalg.inv(S)
    np.testing.assert_allclose(f.P, P - K @ S @ K.T, rtol=1e-11, atol=1e-15)
    expect = o.boxplus(lay, mu, K @ z[rows])
    np.testing.assert_allclose(o.boxminus(lay, f.mean, expect), 0, atol=1e-12)
    # had row 2 survived, clone-2 x would have moved: it must not
    assert abs(f.mean[20]) < 1e-12


def test_all_features_rejected_skips_update():
    # Msckf.hpp:250: innovation.rows() == 0 -> no correction at all
    k = 2
    lay = o.layout(o.MULTI, k)
    mu = o.identity_state(lay)
    P = 0.01 * np.eye(o.dof(lay))
    f = o.Msckf(k, mu, P)
    st, no = f.update([9.0, 9.0, -9.0, 9.0], o.mm_python(lambda X: np.array([X[13], X[14], X[20], X[21]])),
                      0.01 * np.eye(4))
    assert st == 0 and no == 2
    np.testing.assert_array_equal(f.mean, mu)
    np.testing.assert_array_equal(f.P, P)


def test_msckf_predict_leaves_cross_covariance_stale():
    # Appendix B.3 / Msckf.hpp:171-182 (commented out in the reference)
    s = sc.synthetic_msckf(1, 2, m=4)
    f = o.Msckf(2, s["mean"][0], s["P"][0])
    u = s["u"][0]
    f.predict(o.pm_delta_pose(u[0:3], u[3:7], u[7:10], u[10:13]), s["Q"])
    P1 = f.P
    np.testing.assert_array_equal(P1[12:, :], s["P"][0][12:, :])
    np.testing.assert_array_equal(P1[:12, 12:], s["P"][0][:12, 12:])
    assert np.abs(P1[:12, :12] - s["P"][0][:12, :12]).max() > 1e-4


def test_usckf_ctor_cloning_structure():
    # Usckf.hpp:90-103, :391-433: [[P,P,0],[P,P,P],[0,P,P]] -- indefinite (Appendix B.1)
    u = sc.usckf_unit_test()
    f = o.Usckf(state13=u["state_single"], P0_12=u["P0_single"])
    P, B = f.P, u["P0_single"]
    Z = np.zeros((12, 12))
    np.testing.assert_array_equal(P, np.block([[B, B, Z], [B, B, B], [Z, B, B]]))
    assert np.linalg.eigvalsh(P).min() < 0


def test_usckf_set_measurement_wipes_cross_terms():
    # Usckf.hpp:322-389
    u = sc.usckf_unit_test()
    f = o.Usckf(state13=u["state_single"], P0_12=u["P0_single"])
    P36 = f.P
    sizes = []
    for mode, z, R in u["set_measurements"]:
        f.set_measurement(mode, z, R)
        sizes.append(f.N)
    assert sizes == [39, 48, 48]                            # UsckfUnitTest.cpp:210,216,225
    P = f.P
    np.testing.assert_array_equal(P[:36, :36], P36)
    np.testing.assert_array_equal(P[:36, 36:], 0)
    np.testing.assert_array_equal(P[36:39, 36:39], 0.05 * np.eye(3))
    np.testing.assert_array_equal(P[39:, 39:], 0.008 * np.eye(9))
    np.testing.assert_array_equal(f.mean[39:42], [3.35] * 3)
    np.testing.assert_array_equal(f.mean[42:], [1.34] * 9)


def test_usckf_literal_update_reports_llt_failure():
    # Appendix B.1: the literal USCKF_DYNAMIC update() factors an indefinite 48x48 matrix; the
    # reference never checks LLT::info().  The oracle reports it instead of mimicking Eigen's blocking.
    u = sc.usckf_unit_test()
    f = o.Usckf(state13=u["state_single"], P0_12=u["P0_single"])
    for mode, z, R in u["set_measurements"]:
        f.set_measurement(mode, z, R)
    pm = o.pm_const_velocity(u["velocity"], u["angular_velocity"], u["dt"])
    for _ in range(u["n_predict"]):
        assert f.predict(pm, u["Q"]) == 0
    st, _ = f.update(u["z"], o.mm_vo_relative(), u["R"])
    assert st & o.LLT_FAIL


def test_host_functor_equals_builtin_model():
    # Tier B (opaque functor, the reference's boost::bind form) == Tier A (registered model)
    from oracle import np_check as npc
    s = sc.synthetic_msckf(1, 1, m=2)
    u = s["u"][0]
    a = o.Msckf(1, s["mean"][0], s["P"][0])
    b = o.Msckf(1, s["mean"][0], s["P"][0])
    a.predict(o.pm_delta_pose(u[0:3], u[3:7], u[7:10], u[10:13]), s["Q"])
    b.predict(o.pm_python(lambda x: npc.pm_delta_pose(x, u[0:3], u[3:7], u[7:10], u[10:13])), s["Q"])
    np.testing.assert_allclose(a.P, b.P, rtol=1e-12)
    np.testing.assert_allclose(o.boxminus(a.lay, a.mean, b.mean), 0, atol=1e-13)
