"""GPU parity tests: the HIP path (through the C ABI, include/slk.h) against the CPU oracle, the
committed golden fixtures and closed-form answers.  Tolerance: north_star asks for 1e-6 relative
on state and covariance; these tests hold the fp64 kernels to TOL = 1e-9 (observed ~1e-13).
Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest

from oracle import oracle as o
import scenarios as sc

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-9


@pytest.fixture(scope="module")
def slk():
    import torch  # noqa: F401  (loads the HIP runtime the library binds to)
    from slkpkg import slk as mod
    assert mod.device_count() > 0, "no MI355X visible"
    return mod


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def mean_err(lay, a, b):
    return float(np.abs(o.boxminus(lay, a, b)).max())


def test_mfma_f64_fragment_layout(slk):
    assert slk.load_library().slk_selftest_mfma(0) == 0


# ------------------------------------------------------------------ Msckf
@pytest.mark.parametrize("k", [0, 1, 4, 8, 31])
def test_msckf_unit_test_scenario_against_golden(slk, k):
    g = np.load(os.path.join(G, "msckf_unit_test.npz"))
    t = sc.msckf_unit_test(k)
    lay = o.layout(o.MULTI, k)
    f = slk.Msckf(t["mean"], t["P"])
    u = np.r_[t["dpos"], t["dquat"], t["velocity"], t["angular_velocity"]]
    for i in range(t["n_predict"]):
        f.predict(slk.PM_DELTA_POSE, u, t["Q"])
        assert rel(f.getPk()[0], g[f"k{k}_pred{i}_P"]) <= TOL
        assert mean_err(lay, f.muState()[0], g[f"k{k}_pred{i}_mean"]) <= TOL
    z, feat = g[f"k{k}_z"], g[f"k{k}_feat"]
    f.update(z[None, :], slk.MM_FEATURE_PROJ, feat.reshape(1, -1), 0.01 * np.eye(z.size))
    assert f.status()[0] == 0
    assert f.outliers()[0] == int(g[f"k{k}_outliers"][0])
    assert rel(f.getPk()[0], g[f"k{k}_upd_P"]) <= TOL
    assert mean_err(lay, f.muState()[0], g[f"k{k}_upd_mean"]) <= TOL


def test_msckf_batch_golden_with_outliers(slk):
    g = np.load(os.path.join(G, "msckf_batch.npz"))
    s = sc.synthetic_msckf(8, 8)
    lay = o.layout(o.MULTI, 8)
    f = slk.Msckf(s["mean"], s["P"])
    tot = np.zeros(8, dtype=np.int64)
    for _ in range(3):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], g["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        tot += f.outliers()
    st = f.status()
    np.testing.assert_array_equal(tot, g["outliers"])
    assert st[3] == slk.ST_ALL_REJECTED and (np.delete(st, 3) == 0).all()
    P, M = f.getPk(), f.muState()
    for b in range(8):
        assert rel(P[b], g["P"][b]) <= TOL, b
        assert mean_err(lay, M[b], g["mean"][b]) <= TOL, b


@pytest.mark.parametrize("k,m,B", [(0, 2, 96), (1, 2, 96), (3, 6, 64), (8, 8, 64), (10, 8, 8), (12, 8, 16), (13, 8, 6),
                                   (14, 8, 6), (15, 8, 6), (19, 8, 4), (24, 8, 4), (31, 8, 5),
                                   # one tile row per wave (N = 36 .. 60) with fewer than 8 rows: the factor-update path
                                   (4, 2, 16), (5, 4, 16), (6, 6, 16), (7, 2, 8), (8, 4, 16), (8, 6, 16),
                                   # exact-shape instantiations (m = 8): k = 4 .. 7
                                   (4, 8, 16), (5, 8, 16), (6, 8, 16), (7, 8, 16),
                                   # large-state factor update / odd-even rebuild with fewer than 8 rows
                                   (9, 2, 4), (12, 4, 4), (16, 6, 4), (20, 2, 3), (31, 4, 2), (32, 6, 2)])
def test_msckf_step_against_oracle(slk, k, m, B):
    s = sc.synthetic_msckf(B, k, m=m, seed=100 + k)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    steps = 3
    f = slk.Msckf(s["mean"], s["P"])
    tot = np.zeros(B, dtype=np.int64)
    for _ in range(steps):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        tot += f.outliers()
    mean, P = s["mean"].copy(), s["P"].copy()
    st, out = o.msckf_step_batch(k, m, steps, mean, P, s["u"], s["feat"], s["z"], s["Q"], s["R"])
    assert st == 0 and (f.status() & ~slk.ST_ALL_REJECTED == 0).all()
    np.testing.assert_array_equal(tot, out)
    Pg, Mg = f.getPk(), f.muState()
    for b in range(B):
        assert rel(Pg[b], P[b].reshape(N, N).T) <= TOL, b
        assert mean_err(lay, Mg[b], mean[b]) <= TOL, b


@pytest.mark.parametrize("k,B", [(8, 12), (12, 6), (19, 4), (31, 3)])
def test_msckf_rejected_rows_in_the_factor_update(slk, k, B):
    # a feature pushed out of the gate: its two rows drop out of the factor update (identity row / column of the prefix
    # matrices) -- the N = 60 kernel and the large-state path against the oracle, outlier counts included
    m = 8
    s = sc.synthetic_msckf(B, k, m=m, seed=900 + k)
    z = s["z"].copy()
    z[::2, 2:4] += 0.8                       # every other filter: feature 1 far off
    z[1::3, 6:8] -= 0.9                      # every third: feature 3 too
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    f = slk.Msckf(s["mean"], s["P"])
    tot = np.zeros(B, dtype=np.int64)
    for _ in range(2):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], z, slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        tot += f.outliers()
    mean, P = s["mean"].copy(), s["P"].copy()
    st, out = o.msckf_step_batch(k, m, 2, mean, P, s["u"], s["feat"], z, s["Q"], s["R"])
    assert st == 0 and (f.status() & ~slk.ST_ALL_REJECTED == 0).all()
    np.testing.assert_array_equal(tot, out)
    assert tot.sum() > 0
    Pg, Mg = f.getPk(), f.muState()
    for b in range(B):
        assert rel(Pg[b], P[b].reshape(N, N).T) <= TOL, b
        assert mean_err(lay, Mg[b], mean[b]) <= TOL, b


@pytest.mark.parametrize("k,m,B", [(2, 4, 4), (8, 8, 4), (12, 8, 3), (31, 8, 2)])
def test_msckf_update_with_a_wrapped_rotation_column(slk, k, m, B):
    # a rotation variance beyond pi^2: a column of the factor is longer than pi, log(exp(v)) wraps (MTK's atan form) and
    # covXZ is no longer L A -- the kernels leave the factor-update path for the plain one (large states: the blocked
    # factorisation of the downdated matrix)
    s = sc.synthetic_msckf(B, k, m=m, seed=1200 + k)
    N = s["N"]
    P = s["P"].copy().reshape(B, N, N)
    P[:, 4, 4] += 11.0
    P = np.ascontiguousarray(P)
    lay = o.layout(o.MULTI, k)
    f = slk.Msckf(s["mean"], P)
    f.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"], gate=0)
    Pg, Mg = f.getPk(), f.muState()
    for b in range(B):
        r = o.Msckf(k, s["mean"][b], P[b])
        st, _ = r.update(s["z"][b], o.mm_feature_proj(s["feat"][b]), s["R"], gate=False)
        assert st == 0 and f.status()[b] == 0
        assert rel(Pg[b], r.P) <= TOL and mean_err(lay, Mg[b], r.mean) <= TOL


@pytest.mark.parametrize("k,var", [(8, 2.0), (8, 0.6), (5, 2.5), (4, 1.4)])
def test_msckf_update_with_rotation_columns_beyond_one_radian(slk, k, var):
    # rotation variances of 0.6 .. 2.5 rad^2: columns of the factor between ~0.8 and 1.6 rad -- below pi (covXZ = L A still
    # holds) but around / beyond the 1 rad domain of the exp / log series of the exact-shape fast path (k = 4 .. 8, m = 8),
    # which must hand such filters to the general body (libm routes) before it has written anything; mixed batch: the
    # filters of even index keep the synthetic covariance and stay on the fast path
    B, m = 6, 8
    s = sc.synthetic_msckf(B, k, m=m, seed=1250 + k)
    N = s["N"]
    P = s["P"].copy().reshape(B, N, N)
    for b in range(1, B, 2):
        P[b, 3, 3] += var
        P[b, 12 + 6 * (k - 1) + 4, 12 + 6 * (k - 1) + 4] += 0.5 * var
    P = np.ascontiguousarray(P)
    lay = o.layout(o.MULTI, k)
    f = slk.Msckf(s["mean"], P)
    f.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"], gate=0)
    Pg, Mg = f.getPk(), f.muState()
    for b in range(B):
        r = o.Msckf(k, s["mean"][b], P[b])
        st, _ = r.update(s["z"][b], o.mm_feature_proj(s["feat"][b]), s["R"], gate=False)
        assert st == 0 and f.status()[b] == 0
        assert rel(Pg[b], r.P) <= TOL and mean_err(lay, Mg[b], r.mean) <= TOL, b


@pytest.mark.parametrize("k,m,B", [(2, 4, 3), (8, 8, 3), (12, 8, 3), (31, 8, 2)])
def test_msckf_update_with_a_non_spd_innovation_covariance(slk, k, m, B):
    # R = -0.3 I makes S = cov(Z) + R indefinite: the reference inverts it with PartialPivLU all the same (Msckf.hpp:257);
    # the kernels leave the Cholesky-based gain and the factor update for Gauss-Jordan + a fresh factorisation
    s = sc.synthetic_msckf(B, k, m=m, seed=1300 + k)
    N = s["N"]
    R = -0.3 * np.eye(m)
    lay = o.layout(o.MULTI, k)
    f = slk.Msckf(s["mean"], s["P"])
    f.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], R, gate=0)
    Pg, Mg = f.getPk(), f.muState()
    for b in range(B):
        r = o.Msckf(k, s["mean"][b], s["P"][b].reshape(N, N))
        st, _ = r.update(s["z"][b], o.mm_feature_proj(s["feat"][b]), R, gate=False)
        assert st == 0 and f.status()[b] == 0
        assert rel(Pg[b], r.P) <= TOL and mean_err(lay, Mg[b], r.mean) <= TOL


def test_separate_predict_update_equals_fused_step(slk):
    s = sc.synthetic_msckf(16, 4, m=8, seed=7)
    a = slk.Msckf(s["mean"], s["P"])
    b = slk.Msckf(s["mean"], s["P"])
    a.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    b.predict(slk.PM_DELTA_POSE, s["u"], s["Q"])
    b.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    np.testing.assert_array_equal(a.getPk(), b.getPk())
    np.testing.assert_array_equal(a.muState(), b.muState())


def test_tier_b_host_functor_equals_registered_model(slk):
    from oracle import np_check as npc
    s = sc.synthetic_msckf(3, 2, m=4, seed=8)
    lay = o.layout(o.MULTI, 2)
    a = slk.Msckf(s["mean"], s["P"])
    b = slk.Msckf(s["mean"], s["P"])
    a.predict(slk.PM_DELTA_POSE, s["u"], s["Q"])
    a.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    # the reference's form: opaque callables (boost::bind), here per filter through the sigma-point API
    X = b.predict_sigma_points()
    Y = np.array([[npc.pm_delta_pose(x, s["u"][i, 0:3], s["u"][i, 3:7], s["u"][i, 7:10], s["u"][i, 10:13]) for x in X[i]]
                  for i in range(3)])
    lib = slk.load_library()
    Qc = np.ascontiguousarray(s["Q"].T)
    assert lib.slk_predict_from_sigma(b._h, Y.ctypes.data, Qc.ctypes.data, 0, slk.HOST) == 0
    X = b.update_sigma_points()
    Z = np.ascontiguousarray([[npc.mm_feature_proj(x, s["feat"][i]) for x in X[i]] for i in range(3)])
    Rc = np.ascontiguousarray(s["R"].T)
    z = np.ascontiguousarray(s["z"])
    assert lib.slk_update_from_sigma(b._h, Z.ctypes.data, z.ctypes.data, 4, Rc.ctypes.data, 0, 1, slk.HOST) == 0
    for i in range(3):
        assert rel(b.getPk()[i], a.getPk()[i]) <= TOL
        assert mean_err(lay, b.muState()[i], a.muState()[i]) <= TOL


def test_linear_kat_closed_form_on_gpu(slk):
    # G3 on the device: position measurement of clone 1, rotations decoupled => closed-form Kalman update
    rng = np.random.default_rng(5)
    k = 2
    lay = o.layout(o.MULTI, k)
    N = o.dof(lay)
    mu = o.set_from_vector(lay, rng.normal(size=N))
    vec_idx = [i for i in range(N) if not (3 <= i < 6 or (i >= 12 and (i - 12) % 6 >= 3))]
    rot_idx = [i for i in range(N) if i not in vec_idx]

    def spd(n, scale):
        A = rng.normal(0, scale, (n, n))
        return A @ A.T + 0.01 * np.eye(n)

    P = np.zeros((N, N))
    P[np.ix_(vec_idx, vec_idx)] = spd(len(vec_idx), 0.05)
    P[np.ix_(rot_idx, rot_idx)] = spd(len(rot_idx), 0.02)
    R = 0.02 * np.eye(3)
    H = np.zeros((3, N))
    H[:, 12:15] = np.eye(3)
    z = mu[13:16] + np.array([0.05, -0.02, 0.01])
    f = slk.Msckf(mu, P)
    f.update(z[None, :], slk.MM_POSE_POSITION, np.array([1.0]), R, gate=0)
    S = H @ P @ H.T + R
    K = P @ H.T @ np.linalg.inv(S)
    assert rel(f.getPk()[0], P - K @ S @ K.T) <= 1e-10
    assert mean_err(lay, f.muState()[0], o.boxplus(lay, mu, K @ (z - mu[13:16]))) <= 1e-11


def test_all_rejected_keeps_predicted_state(slk):
    s = sc.synthetic_msckf(4, 2, m=4, seed=9)
    z = s["z"] + 50.0
    a = slk.Msckf(s["mean"], s["P"])
    b = slk.Msckf(s["mean"], s["P"])
    a.step(slk.PM_DELTA_POSE, s["u"], s["Q"], z, slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    b.predict(slk.PM_DELTA_POSE, s["u"], s["Q"])
    assert (a.status() == slk.ST_ALL_REJECTED).all() and (a.outliers() == 2).all()
    np.testing.assert_array_equal(a.getPk(), b.getPk())
    np.testing.assert_array_equal(a.muState(), b.muState())


def test_llt_failure_is_reported_and_state_kept(slk):
    s = sc.synthetic_msckf(2, 1, m=2, seed=10)
    P = s["P"].copy()
    P[1] = -P[1]                                   # filter 1: not positive definite
    f = slk.Msckf(s["mean"], P)
    f.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    st = f.status()
    assert st[0] == 0 and st[1] & slk.ST_LLT_FAIL
    np.testing.assert_array_equal(f.getPk()[1], P[1])
    np.testing.assert_array_equal(f.muState()[1], s["mean"][1])


@pytest.mark.parametrize("k", [0, 1, 4, 8, 12, 31])
@pytest.mark.parametrize("how", ["negated", "late pivot", "nan"])
def test_llt_failure_in_every_factor_kernel(slk, k, how):
    """Msckf.hpp:407-413 (Eigen::LLT, info() never read): every factorisation of this library -- the register Cholesky by rows
    (N = 12 / 18), the one-wave factor kernel by panel rows (N = 36 / 60), the workgroup kernels (N = 84) and the
    global-workspace one (N = 198) -- reports a covariance that is not positive definite and leaves that filter alone:
    a negated matrix (first pivot), an indefinite one whose LAST pivot fails, a NaN on the diagonal; update and step."""
    m = 3 if k == 0 else 8
    s = sc.synthetic_msckf(3, k, m=(2 if k == 0 else m), seed=77 + k)
    N = 12 + 6 * k
    P = s["P"].copy()
    if how == "negated":
        P[1] = -P[1]
    elif how == "late pivot":
        P[1, N - 1, N - 1] = -1.0
    else:
        P[1, N // 2, N // 2] = np.nan

    def update(f):
        if k == 0:
            f.update(s["mean"][:, 0:3] + 0.05, slk.MM_POSE_POSITION, np.array([0.0]), 0.01 * np.eye(3), gate=0)
        else:
            f.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])

    f, good = slk.Msckf(s["mean"], P), slk.Msckf(s["mean"], s["P"])
    update(f)
    update(good)
    st = f.status()
    assert st[0] == 0 and st[2] == 0 and st[1] & slk.ST_LLT_FAIL
    got = f.getPk()
    np.testing.assert_array_equal(got[1], P[1])                       # (NaN == NaN position-wise)
    np.testing.assert_array_equal(f.muState()[1], s["mean"][1])
    np.testing.assert_array_equal(got[[0, 2]], good.getPk()[[0, 2]])   # the neighbours are not disturbed
    if how != "late pivot" or k == 0:              # predict factors the 12 x 12 state block only
        g = slk.Msckf(s["mean"], P)
        g.predict(slk.PM_DELTA_POSE, s["u"], s["Q"])
        bad12 = how == "negated" or (how == "nan" and N // 2 < 12) or (how == "late pivot" and k == 0)
        assert bool(g.status()[1] & slk.ST_LLT_FAIL) == bad12
        if bad12:
            np.testing.assert_array_equal(g.getPk()[1], P[1])


def test_api_misuse_is_rejected(slk):
    s = sc.synthetic_msckf(2, 1, m=2, seed=11)
    f = slk.Msckf(s["mean"], s["P"])
    lib = slk.load_library()
    z = np.zeros((2, 3))
    R = np.eye(3)
    assert lib.slk_update(f._h, slk.MM_FEATURE_PROJ, None, 0, z.ctypes.data, 3, R.ctypes.data, 0, 1, slk.HOST) == slk.E_INVALID
    assert lib.slk_update(f._h, 77, None, 0, z.ctypes.data, 2, R.ctypes.data, 0, 1, slk.HOST) == slk.E_INVALID
    assert lib.slk_predict(f._h, slk.PM_DELTA_POSE, None, 0, None, 0, slk.HOST) == slk.E_INVALID


@pytest.mark.parametrize("k,m,B", [(35, 8, 2), (40, 4, 2), (49, 8, 1)])
def test_msckf_windows_beyond_the_lds_kernels(slk, k, m, B):
    # N = 222 / 252 / 306: the reference's MultiState is unbounded (State.hpp:342, :373-376); these windows run the plain
    # global-workspace kernel (csrc/slk_general.hpp) -- predict + update with the gate, a second step, then the Tier-B
    # functor path (sigma points out, Z back) and checkSigmaPoints, all against the oracle
    s = sc.synthetic_msckf(B, k, m=m, seed=1700 + k)
    N = s["N"]
    lay = o.layout(o.MULTI, k)
    f = slk.Msckf(s["mean"], s["P"])
    ref = [o.Msckf(k, s["mean"][b], s["P"][b].reshape(N, N)) for b in range(B)]
    z = s["z"].copy()
    z[0, 0] += 4.0                                             # one gross outlier block in filter 0
    for step in range(2):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], z, slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        out = f.outliers()
        Pg, Mg = f.getPk(), f.muState()
        assert (f.status() == 0).all()
        for b in range(B):
            u = s["u"][b]
            assert ref[b].predict(o.pm_delta_pose(u[0:3], u[3:7], u[7:10], u[10:13]), s["Q"]) == 0
            st, no = ref[b].update(z[b], o.mm_feature_proj(s["feat"][b]), s["R"])
            assert st == 0 and no == out[b]
            assert rel(Pg[b], ref[b].P) <= TOL and mean_err(lay, Mg[b], ref[b].mean) <= TOL, (step, b)
    assert out[0] >= 1
    # opaque functor path (filter 0): the same model evaluated on the host from the emitted sigma points
    g = slk.Msckf(f.muState()[:1], f.getPk()[:1])
    g.update_functor(s["z"][:1], lambda x: _proj(s["feat"][0], x, k), s["R"], gate=0)
    f.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"], gate=0)
    assert rel(g.getPk()[0], f.getPk()[0]) <= 1e-11 and mean_err(lay, g.muState()[0], f.muState()[0]) <= 1e-11
    cov_err, mean_err_ = f.checkSigmaPoints()
    assert cov_err.max() <= 1e-9 and mean_err_.max() <= 1e-9


def _proj(feat, X, k):
    """Feature projection of one sigma point (numpy): landmark in the frame of the observing pose, normalised image point."""
    z = []
    for fx, fy, fz, c in feat.reshape(-1, 4):
        c = int(c)
        sp = 0 if c == 0 else 13 + 7 * (c - 1)
        p, q = X[sp:sp + 3], X[sp + 3:sp + 7]
        qc = np.array([-q[0], -q[1], -q[2], q[3]])
        l = sc.quat_rotate(qc, np.array([fx, fy, fz]) - p)
        z += [l[0] / l[2], l[1] / l[2]]
    return np.array(z)


# ------------------------------------------------------------------ full-size properties (BASELINE cfg3)
def test_full_size_batch_properties_and_sampled_parity(slk):
    B, k, m = 4096, 8, 8
    s = sc.synthetic_msckf(B, k, m=m, seed=0x5EED0000)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    f = slk.Msckf(s["mean"], s["P"])
    steps = 2
    tot = np.zeros(B, dtype=np.int64)
    for _ in range(steps):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        tot += f.outliers()
    P, M = f.getPk(), f.muState()
    st = f.status()
    assert (st & ~slk.ST_ALL_REJECTED == 0).all()
    # size-independent properties: exact symmetry of the rebuilt covariance, positive definiteness,
    # unit quaternions, finite values
    upd = st == 0
    assert np.isfinite(P).all() and np.isfinite(M).all()
    np.testing.assert_array_equal(P[upd], np.transpose(P[upd], (0, 2, 1)))
    assert np.linalg.eigvalsh(P).min() > 0
    q = np.concatenate([M[:, 3:7][:, None, :]] + [M[:, 13 + 7 * c + 3:13 + 7 * c + 7][:, None, :] for c in range(k)], axis=1)
    np.testing.assert_allclose(np.linalg.norm(q, axis=-1), 1.0, atol=1e-12)
    # sampled parity against the oracle
    idx = np.r_[0:16, B - 16:B]
    mean, Pc = s["mean"][idx].copy(), s["P"][idx].copy()
    stc, out = o.msckf_step_batch(k, m, steps, mean, Pc, np.ascontiguousarray(s["u"][idx]),
                                  np.ascontiguousarray(s["feat"][idx]), np.ascontiguousarray(s["z"][idx]), s["Q"], s["R"])
    assert stc == 0
    np.testing.assert_array_equal(tot[idx], out)
    for j, b in enumerate(idx):
        assert rel(P[b], Pc[j].reshape(N, N).T) <= TOL
        assert mean_err(lay, M[b], mean[j]) <= TOL


# ------------------------------------------------------------------ Usckf
def test_usckf_unit_test_scenario_against_golden(slk):
    g = np.load(os.path.join(G, "usckf_unit_test.npz"))
    u = sc.usckf_unit_test()
    f = slk.Usckf(state_single=u["state_single"], P0_single=u["P0_single"])
    np.testing.assert_array_equal(f.PkAugmentedState()[0], g["ctor_P"])
    np.testing.assert_array_equal(f.muState()[0], g["ctor_mean"])
    for i, (mode, z, R) in enumerate(u["set_measurements"]):
        f.setMeasurement(mode, z, R)
        np.testing.assert_array_equal(f.PkAugmentedState()[0], g[f"setm{i}_P"])
        np.testing.assert_array_equal(f.muState()[0], g[f"setm{i}_mean"])
    assert f.N == 48
    lay = o.layout(o.AUGMENTED, 0, 3, 9)
    uu = np.r_[u["velocity"], u["angular_velocity"], u["dt"]]
    for i in range(u["n_predict"]):
        f.predict(slk.PM_CONST_VELOCITY, uu, u["Q"])
        assert f.status()[0] == 0
        assert rel(f.PkAugmentedState()[0], g[f"pred{i}_P"]) <= TOL
        assert mean_err(lay, f.muState()[0], g[f"pred{i}_mean"]) <= TOL
    # literal update(): indefinite 48x48 covariance -> reported, state kept
    before_P, before_m = f.PkAugmentedState(), f.muState()
    f.update(u["z"][None, :], slk.MM_VO_RELATIVE, None, u["R"])
    assert f.status()[0] & slk.ST_LLT_FAIL
    np.testing.assert_array_equal(f.PkAugmentedState(), before_P)
    np.testing.assert_array_equal(f.muState(), before_m)


def test_usckf_spd_predict_update_against_golden_and_oracle(slk):
    g = np.load(os.path.join(G, "usckf_spd.npz"))
    s = sc.synthetic_usckf(4)
    lay = o.layout(o.AUGMENTED, 0, 3, 9)
    f = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    h = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    for _ in range(2):
        f.predict(slk.PM_CONST_VELOCITY, s["u"], s["Q"])
        f.update(s["z"], slk.MM_VO_RELATIVE, None, s["R"])
        h.step(slk.PM_CONST_VELOCITY, s["u"], s["Q"], s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    assert (f.status() == 0).all()
    P, M = f.PkAugmentedState(), f.muState()
    for b in range(4):
        assert rel(P[b], g["P"][b]) <= TOL
        assert mean_err(lay, M[b], g["mean"][b]) <= TOL
    np.testing.assert_array_equal(h.PkAugmentedState(), P)
    np.testing.assert_array_equal(h.muState(), M)


def test_usckf_lower_triangle_steady_state_and_completion_on_demand(slk):
    """The unit-test shape keeps only the lower triangle of the covariance up to date between steps (predict, the in-kernel
    factorisation and the exact-shape update read nothing else, Usckf.hpp:537); whatever hands the matrix out or copies
    blocks of it completes the strict upper triangle first."""
    s = sc.synthetic_usckf(4)
    f = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    for _ in range(3):
        f.step(slk.PM_CONST_VELOCITY, s["u"], s["Q"], s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    assert (f.status() == 0).all()
    P = f.PkAugmentedState()
    assert np.abs(P - np.transpose(P, (0, 2, 1))).max() == 0.0
    om, oP = s["mean"].copy(), np.ascontiguousarray(np.transpose(s["P"], (0, 2, 1))).reshape(4, -1)
    assert o.usckf_step_batch(3, 9, 3, om, oP, s["u"], s["z"], s["Q"], s["R"]) == 0
    for b in range(4):
        assert rel(P[b], oP[b].reshape(48, 48).T) <= TOL
    # cloning right after the steps (copies blocks of both triangles) == cloning of a handle that was given the whole matrix
    g = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    for _ in range(3):
        g.step(slk.PM_CONST_VELOCITY, s["u"], s["Q"], s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    g.cloning(slk.STATEK_I)
    h = slk.Usckf(mean=f.muState(), P=P, nfk=3, nfkl=9)
    h.cloning(slk.STATEK_I)
    np.testing.assert_array_equal(g.PkAugmentedState(), h.PkAugmentedState())
    # a different measurement model after lower-triangle steps (the general kernel stages the whole matrix)
    g2 = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    h2 = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    g2.step(slk.PM_CONST_VELOCITY, s["u"], s["Q"], s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    h2.step(slk.PM_CONST_VELOCITY, s["u"], s["Q"], s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    h2 = slk.Usckf(mean=h2.muState(), P=h2.PkAugmentedState(), nfk=3, nfkl=9)        # (the whole matrix, through the host)
    X = g2.update_sigma_points()
    np.testing.assert_array_equal(X, h2.update_sigma_points())


@pytest.mark.parametrize("nfk,nfkl", [(3, 0), (3, 18), (3, 23)])
def test_usckf_other_feature_counts_against_oracle(slk, nfk, nfkl):
    # N = 39 and 57 take the three-launch path (predict / factor / update), N = 62 the fused kernel
    s = sc.synthetic_usckf(3, nfk=nfk, nfkl=nfkl, seed=400 + nfkl)
    lay = o.layout(o.AUGMENTED, 0, nfk, nfkl)
    f = slk.Usckf(mean=s["mean"], P=s["P"], nfk=nfk, nfkl=nfkl)
    f.predict(slk.PM_CONST_VELOCITY, s["u"], s["Q"])
    f.update(s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    f.step(slk.PM_CONST_VELOCITY, s["u"], s["Q"], s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    assert (f.status() == 0).all()
    P, M = f.PkAugmentedState(), f.muState()
    for b in range(3):
        g = o.Usckf(nfk=nfk, nfkl=nfkl, mean=s["mean"][b], P=s["P"][b])
        for _ in range(2):
            assert g.predict(o.pm_const_velocity(s["u"][b, 0:3], s["u"][b, 3:6], s["u"][b, 6]), s["Q"]) == 0
            stc, acc = g.update(s["z"][b], o.mm_vo_relative(), s["R"])
            assert stc == 0 and acc == 1
        assert rel(P[b], g.P) <= TOL and mean_err(lay, M[b], g.mean) <= TOL


def test_usckf_whole_vector_gate_and_other_models(slk):
    # Usckf.hpp:262-302 with a chi-square gate (mt) instead of accept_any: accepted and rejected filters
    s = sc.synthetic_usckf(4, seed=77)
    z = s["z"].copy()
    z[1] += 5.0                                     # filter 1: gross innovation -> rejected by chi2(3)
    f = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    f.update(z, slk.MM_VO_RELATIVE, None, s["R"], gate=3)
    st, out = f.status(), f.outliers()
    assert out[1] == 1 and st[1] == slk.ST_ALL_REJECTED and (np.delete(out, 1) == 0).all()
    lay = o.layout(o.AUGMENTED, 0, 3, 9)
    P, M = f.PkAugmentedState(), f.muState()
    for b in range(4):
        g = o.Usckf(nfk=3, nfkl=9, mean=s["mean"][b], P=s["P"][b])
        stc, acc = g.update(z[b], o.mm_vo_relative(), s["R"], gate_dof=3)
        assert stc == 0 and acc == (0 if b == 1 else 1)
        assert rel(P[b], g.P) <= TOL and mean_err(lay, M[b], g.mean) <= TOL
    # feature-projection model on the Usckf layout (pose index = statek / statek_l / statek_i)
    feat = np.zeros((4, 2, 4))
    for b in range(4):
        for j, c in enumerate((0, 2)):
            p = s["mean"][b, 13 * c:13 * c + 3]
            q = s["mean"][b, 13 * c + 3:13 * c + 7]
            feat[b, j, 0:3] = p + sc.quat_rotate(q, np.array([0.3 * (j + 1), -0.2, 5.0 + j]))
            feat[b, j, 3] = c
    zz = np.tile(np.array([0.06, -0.04, 0.1, -0.03]), (4, 1))
    f2 = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    f2.update(zz, slk.MM_FEATURE_PROJ, feat, 0.01 * np.eye(4))
    assert (f2.status() == 0).all()
    P2, M2 = f2.PkAugmentedState(), f2.muState()
    for b in range(4):
        g = o.Usckf(nfk=3, nfkl=9, mean=s["mean"][b], P=s["P"][b])
        stc, acc = g.update(zz[b], o.mm_feature_proj(feat[b]), 0.01 * np.eye(4))
        assert stc == 0 and acc == 1
        assert rel(P2[b], g.P) <= TOL and mean_err(lay, M2[b], g.mean) <= TOL


def test_usckf_functor_path_equals_registered_model(slk):
    from oracle import np_check as npc
    s = sc.synthetic_usckf(2, seed=78)
    lay = o.layout(o.AUGMENTED, 0, 3, 9)
    a = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    b = slk.Usckf(mean=s["mean"], P=s["P"], nfk=3, nfkl=9)
    a.predict(slk.PM_CONST_VELOCITY, s["u"], s["Q"])
    a.update(s["z"], slk.MM_VO_RELATIVE, None, s["R"])
    lib = slk.load_library()
    X = b.predict_sigma_points()
    Y = np.ascontiguousarray([[npc.pm_const_velocity(x, s["u"][i, 0:3], s["u"][i, 3:6], s["u"][i, 6]) for x in X[i]]
                              for i in range(2)])
    Qc = np.ascontiguousarray(s["Q"].T)
    assert lib.slk_predict_from_sigma(b._h, Y.ctypes.data, Qc.ctypes.data, 0, slk.HOST) == 0
    b.update_functor(s["z"], lambda x: npc.mm_vo_relative(x, 3), s["R"])
    for i in range(2):
        assert rel(b.PkAugmentedState()[i], a.PkAugmentedState()[i]) <= TOL
        assert mean_err(lay, b.muState()[i], a.muState()[i]) <= TOL


def test_msckf_window_resize_keeps_working(slk):
    # sliding window: the caller changes the clone set (Msckf.hpp:381-395): resize, upload, step
    s4 = sc.synthetic_msckf(3, 4, m=4, seed=79)
    s6 = sc.synthetic_msckf(3, 6, m=4, seed=80)
    f = slk.Msckf(s4["mean"], s4["P"])
    f.step(slk.PM_DELTA_POSE, s4["u"], s4["Q"], s4["z"], slk.MM_FEATURE_PROJ, s4["feat"], s4["R"])
    assert slk.load_library().slk_msckf_resize(f._h, 6) == 0 and f.N == 48
    f.set_state(s6["mean"], s6["P"])
    f.clear_status()
    f.step(slk.PM_DELTA_POSE, s6["u"], s6["Q"], s6["z"], slk.MM_FEATURE_PROJ, s6["feat"], s6["R"])
    mean, P = s6["mean"].copy(), s6["P"].copy()
    st, out = o.msckf_step_batch(6, 4, 1, mean, P, s6["u"], s6["feat"], s6["z"], s6["Q"], s6["R"])
    lay = o.layout(o.MULTI, 6)
    for b in range(3):
        assert rel(f.getPk()[b], P[b].reshape(48, 48).T) <= TOL
        assert mean_err(lay, f.muState()[b], mean[b]) <= TOL


def test_dead_reckon_delta_and_fused_predict(slk):
    # DeadReckon::updatePose (src/core/DeadReckon.hpp:129-239): batch op and the fused process model
    g = np.load(os.path.join(G, "dead_reckon.npz"))
    u = g["u"]
    s = sc.synthetic_msckf(8, 2, m=2, seed=77)
    lay = o.layout(o.MULTI, 2)
    f = slk.Msckf(s["mean"], s["P"])
    for blk in range(8):                                   # the op works on the handle's batch of 8
        d = f.dead_reckon(u[8 * blk:8 * blk + 8])
        assert np.abs(d - g["delta"][8 * blk:8 * blk + 8]).max() <= 1e-14
    h = slk.Msckf(s["mean"], s["P"])                       # same predicts through the explicit delta poses
    for step in range(2):
        f.predict(slk.PM_DEAD_RECKON, u[8 * step:8 * step + 8], s["Q"])
        h.predict(slk.PM_DELTA_POSE, g["delta"][8 * step:8 * step + 8], s["Q"])
    assert (f.status() == 0).all()
    Pg, Mg, Ph, Mh = f.getPk(), f.muState(), h.getPk(), h.muState()
    for b in range(8):
        assert rel(Pg[b], g["P"][b]) <= TOL and mean_err(lay, Mg[b], g["mean"][b]) <= TOL
        assert rel(Pg[b], Ph[b]) <= 1e-13 and mean_err(lay, Mg[b], Mh[b]) <= 1e-13
    # device-resident inputs and outputs (SLK_DEVICE) without any other runtime: a second handle's buffers are the
    # scratch -- its mean buffer [8][27] carries the inputs (row stride 27), its covariance buffer takes the result
    lib = slk.load_library()
    scratch = slk.Msckf(s["mean"], s["P"])
    rows = np.zeros((8, 27))
    rows[:, :13] = u[:8]
    import ctypes as C
    assert lib.slk_set_state(scratch._h, rows.ctypes.data, None, slk.HOST) == 0
    lib.slk_mean_device_ptr.restype = C.c_void_p
    lib.slk_cov_device_ptr.restype = C.c_void_p
    src, dst = lib.slk_mean_device_ptr(scratch._h), lib.slk_cov_device_ptr(scratch._h)
    assert lib.slk_dead_reckon(f._h, C.c_void_p(src), 27, C.c_void_p(dst), slk.DEVICE) == 0
    f.sync()                                               # the two handles own different streams
    out = np.zeros((8, 24 * 24))
    assert lib.slk_get_state(scratch._h, None, out.ctypes.data, slk.HOST) == 0
    assert np.abs(out.reshape(-1)[:104].reshape(8, 13) - g["delta"][:8]).max() <= 1e-14


@pytest.mark.parametrize("k", [4, 8])
def test_strict_upper_triangle_is_completed_on_demand(slk, k):
    """The exact-shape update kernels (k = 4 .. 8, m = 8) store P+ as lower triangle + diagonal tiles; nothing on the device
    reads more (Msckf.hpp:412, :447 -- Eigen::LLT).  Everything that hands the matrix out completes it first: slk_get_state,
    the zero-copy pointer, the window operations, the EKF update."""
    import torch
    B = 6
    s = sc.synthetic_msckf(B, k, m=8, seed=1234 + k)
    N = 12 + 6 * k

    def stepped(n):
        f = slk.Msckf(s["mean"], s["P"])
        for _ in range(n):
            f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        return f

    f = stepped(3)
    assert (f.status() == 0).all()
    P = f.getPk()
    assert np.abs(P - np.transpose(P, (0, 2, 1))).max() == 0.0            # slk_get_state: exactly symmetric
    # against the oracle (whole matrix)
    for b in range(2):
        r = o.Msckf(k, s["mean"][b], s["P"][b])
        for _ in range(3):
            u = s["u"][b]
            assert r.predict(o.pm_delta_pose(u[0:3], u[3:7], u[7:10], u[10:13]), s["Q"]) == 0
            r.update(s["z"][b], o.mm_feature_proj(s["feat"][b]), s["R"])
        assert np.abs(P[b] - r.P).max() / np.abs(r.P).max() <= 1e-9
    # zero-copy: the pointer call completes the strict upper triangle on the handle's stream
    g = stepped(3)
    _, cov = g.device_pointers()
    g.sync()
    import ctypes as C
    host = np.empty((B, N, N))
    assert torch.cuda.is_available()
    rt = C.CDLL("libamdhip64.so")
    assert rt.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(cov), C.c_size_t(host.nbytes), 2) == 0      # hipMemcpyDeviceToHost
    np.testing.assert_array_equal(np.transpose(host, (0, 2, 1)), P)
    # a window operation right after the steps (whole blocks of P are copied), then the read-out
    h = stepped(3)
    h.clone_pose()
    Ph = h.getPk()
    assert np.abs(Ph - np.transpose(Ph, (0, 2, 1))).max() == 0.0
    np.testing.assert_array_equal(Ph[:, :N, :N], P)


def test_msckf_device_side_window_sliding(slk):
    # clone push / pop on the device (the reference's callers do muState().sensorsk push/pop + setPk,
    # Msckf.hpp:381-395); checked against the same index manipulation in numpy and a following step in the oracle
    B, k, m = 6, 3, 4
    s = sc.synthetic_msckf(B, k, m=m, seed=4242)
    f = slk.Msckf(s["mean"], s["P"])
    N, Nq = s["N"], s["Nq"]
    mean, P = s["mean"].copy(), s["P"].reshape(B, N, N).copy()
    # push: the new clone is the current pose, its rows / columns copy the pose's
    f.clone_pose()
    src = np.concatenate([np.arange(N), np.arange(6)])
    P1 = P[:, src][:, :, src]
    mean1 = np.concatenate([mean, mean[:, 0:7]], axis=1)
    assert f.N == N + 6 and f.Nq == Nq + 7
    assert np.array_equal(f.getPk(), P1) and np.array_equal(f.muState(), mean1)
    # drop the oldest clone
    f.drop_clone(0)
    keep_t = np.concatenate([np.arange(12), np.arange(18, N + 6)])
    keep_s = np.concatenate([np.arange(13), np.arange(20, Nq + 7)])
    P2, mean2 = P1[:, keep_t][:, :, keep_t], mean1[:, keep_s]
    assert f.N == N and np.array_equal(f.getPk(), P2) and np.array_equal(f.muState(), mean2)
    # drop a middle clone, then the window keeps filtering: one step against the oracle
    f.drop_clone(1)
    keep_t = np.concatenate([np.arange(18), np.arange(24, N)])
    keep_s = np.concatenate([np.arange(20), np.arange(27, Nq)])
    P3, mean3 = P2[:, keep_t][:, :, keep_t], mean2[:, keep_s]
    assert np.array_equal(f.getPk(), P3) and np.array_equal(f.muState(), mean3)
    with pytest.raises(slk.SlkError):
        f.drop_clone(5)
    # the pushed clone duplicates the pose, so P3 is singular by construction: regularised (as a caller adding the
    # sensor noise of the new clone would) the window keeps filtering and the step matches the oracle
    k3 = k - 1
    N3 = 12 + 6 * k3
    P3r = P3 + 1e-4 * np.eye(N3)
    f.setPk(P3r)
    feat = s["feat"].copy()
    feat[:, :, 3] = np.minimum(feat[:, :, 3], k3)
    f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, feat, s["R"])
    mo, Po = mean3.copy(), np.ascontiguousarray(P3r).reshape(B, -1)
    st, out = o.msckf_step_batch(k3, m, 1, mo, Po, s["u"], feat, s["z"], s["Q"], s["R"])
    assert st == 0
    np.testing.assert_array_equal(f.outliers(), out)
    lay = o.layout(o.MULTI, k3)
    Pg, Mg = f.getPk(), f.muState()
    for b in range(B):
        assert rel(Pg[b], Po[b].reshape(N3, N3).T) <= TOL and mean_err(lay, Mg[b], mo[b]) <= TOL


@pytest.mark.parametrize("k,m,B", [(8, 20, 8), (8, 32, 8), (4, 12, 1), (10, 16, 4)])
def test_msckf_many_measurement_rows_against_oracle(slk, k, m, B):
    # m > 8 takes the generic row solve, m > 16 the two-tile factorisation of S and the multi-tile S / covXZ path;
    # m = 32 is the largest block the kernels hold on chip (MAXM); B = 1 is the reference's single filter
    s = sc.synthetic_msckf(B, k, m=m, seed=900 + m)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    f = slk.Msckf(s["mean"], s["P"])
    tot = np.zeros(B, dtype=np.int64)
    for _ in range(2):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        tot += f.outliers()
    mean, P = s["mean"].copy(), s["P"].copy()
    st, out = o.msckf_step_batch(k, m, 2, mean, P, s["u"], s["feat"], s["z"], s["Q"], s["R"])
    assert st == 0 and (f.status() & ~slk.ST_ALL_REJECTED == 0).all()
    np.testing.assert_array_equal(tot, out)
    Pg, Mg = f.getPk(), f.muState()
    for b in range(B):
        assert rel(Pg[b], P[b].reshape(N, N).T) <= TOL, b
        assert mean_err(lay, Mg[b], mean[b]) <= TOL, b
    with pytest.raises(slk.SlkError):                        # one row more than the kernels hold
        f.update(np.zeros((B, 34)), slk.MM_FEATURE_PROJ, np.zeros((B, 17, 4)), np.eye(34))


@pytest.mark.parametrize("k,m", [(1, 24), (4, 48), (8, 72), (8, 128)])
def test_msckf_ekf_update_against_golden_and_oracle(slk, k, m):
    # Msckf EKF update (Msckf.hpp:284-349): gate on the unreduced information matrix with the shifted
    # second erase, Householder compression to N rows, gain, covariance, boxplus -- batch of 3 + 5 more filters
    g = np.load(os.path.join(G, "msckf_ekf.npz"))
    e = sc.synthetic_ekf(3, k, m, seed=0xEC0F + k + m)
    lay = o.layout(o.MULTI, k)
    f = slk.Msckf(e["mean"], e["P"])
    f.update_ekf(e["z"], e["zmean"], e["H"], e["R"])
    assert (f.status() == 0).all()
    np.testing.assert_array_equal(f.outliers(), g[f"k{k}_m{m}_outliers"])
    Pg, Mg = f.getPk(), f.muState()
    for b in range(3):
        assert rel(Pg[b], g[f"k{k}_m{m}_P"][b]) <= TOL, b
        assert mean_err(lay, Mg[b], g[f"k{k}_m{m}_mean"][b]) <= TOL, b
    # a second, ungated batch straight against the oracle
    e = sc.synthetic_ekf(5, k, m, seed=4000 + k + m)
    f = slk.Msckf(e["mean"], e["P"])
    f.update_ekf(e["z"], e["zmean"], e["H"], e["R"], gate=False)
    assert (f.status() == 0).all() and (f.outliers() == 0).all()
    Pg, Mg = f.getPk(), f.muState()
    for b in range(5):
        r = o.Msckf(k, e["mean"][b], e["P"][b])
        st, no = r.update_ekf(e["z"][b], e["zmean"][b], e["H"][b], e["R"][b], gate=False)
        assert st == 0 and no == 0
        assert rel(Pg[b], r.P) <= TOL and mean_err(lay, Mg[b], r.mean) <= TOL


@pytest.mark.parametrize("k,m", [(9, 80), (2, 130), (0, 12), (0, 16)])
def test_msckf_ekf_update_other_shapes_against_oracle(slk, k, m):
    # N = 66 > 64 and m = 130 > 128 run the general (global workspace) kernel; N = 12 is the tile kernel's single
    # column-tile case (m = N: every row survives the compression)
    e = sc.synthetic_ekf(4, k, m, seed=7100 + k + m)
    lay = o.layout(o.MULTI, k)
    f = slk.Msckf(e["mean"], e["P"])
    f.update_ekf(e["z"], e["zmean"], e["H"], e["R"], gate=(m > 16))
    Pg, Mg = f.getPk(), f.muState()
    for b in range(4):
        r = o.Msckf(k, e["mean"][b], e["P"][b])
        st, no = r.update_ekf(e["z"][b], e["zmean"][b], e["H"][b], e["R"][b], gate=(m > 16))
        assert st == f.status()[b] and no == f.outliers()[b]
        if st == 0:
            assert rel(Pg[b], r.P) <= TOL and mean_err(lay, Mg[b], r.mean) <= TOL


def test_msckf_ekf_update_edge_cases(slk):
    k, m = 2, 40
    e = sc.synthetic_ekf(4, k, m, seed=515, outliers=False)
    N = e["N"]
    f = slk.Msckf(e["mean"], e["P"])
    with pytest.raises(slk.SlkError):                        # fewer rows than state dimensions (reduceDimension, :806)
        f.update_ekf(e["z"][:, :N - 2], e["zmean"][:, :N - 2], e["H"][:, :N - 2], e["R"][:, :N - 2, :N - 2])
    # most blocks are outliers: fewer than N rows survive -> status bit, filter untouched (the oracle agrees)
    z = e["z"].copy()
    z[:, 0:m - 8] += 60.0
    f.update_ekf(z, e["zmean"], e["H"], e["R"])
    for b in range(4):
        r = o.Msckf(k, e["mean"][b], e["P"][b])
        st, no = r.update_ekf(z[b], e["zmean"][b], e["H"][b], e["R"][b])
        assert st == 16 and no == f.outliers()[b] and f.status()[b] == slk.ST_EKF_ROWS
    assert np.array_equal(f.getPk(), e["P"]) and np.array_equal(f.muState(), e["mean"])
    # an indefinite R makes H P H^T + R non-SPD: reported, filter untouched
    f.clear_status()
    f.update_ekf(e["z"], e["zmean"], e["H"], -np.eye(m))
    assert (f.status() & slk.ST_SINGULAR).all() and np.array_equal(f.getPk(), e["P"])


def test_msckf_long_trajectory_stays_on_the_oracle(slk):
    # 30 fused steps with fresh process inputs and measurements every step (a Monte-Carlo trajectory): the GPU state
    # must track the oracle's without drift beyond the north-star tolerance (observed ~1e-12)
    B, k, m, steps = 12, 8, 8, 30
    s = sc.synthetic_msckf(B, k, m=m, seed=2024)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    rng = np.random.default_rng(5)
    f = slk.Msckf(s["mean"], s["P"])
    mean, P = s["mean"].copy(), s["P"].copy()
    tot_g, tot_o = np.zeros(B, dtype=np.int64), np.zeros(B, dtype=np.int64)
    for t in range(steps):
        u = s["u"].copy()
        u[:, 0:3] += rng.normal(0, 0.02, (B, 3))
        z = s["z"] + rng.normal(0, 0.02, s["z"].shape)
        f.step(slk.PM_DELTA_POSE, u, s["Q"], z, slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        tot_g += f.outliers()
        st, out = o.msckf_step_batch(k, m, 1, mean, P, u, s["feat"], z, s["Q"], s["R"])
        assert st == 0
        tot_o += out
    assert (f.status() & ~slk.ST_ALL_REJECTED == 0).all()
    np.testing.assert_array_equal(tot_g, tot_o)
    Pg, Mg = f.getPk(), f.muState()
    for b in range(B):
        assert rel(Pg[b], P[b].reshape(N, N).T) <= 1e-8, b
        assert mean_err(lay, Mg[b], mean[b]) <= 1e-8, b
