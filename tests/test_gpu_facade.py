"""GPU test of the C++ header facade (include/localization/filters/*.hpp): the reference's unit-test
scenarios compiled as a client program (tests/cpp/facade_scenarios.cpp), checked against the golden
fixtures.  Registered-model path and opaque-functor path must give the same numbers."""
import os

import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-9


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


@pytest.fixture(scope="module")
def res():
    import __graft_entry__ as ge
    ge.build()
    import facade_build
    u = np.load(os.path.join(G, "dead_reckon.npz"))["u"][5]
    return facade_build.run(u)


@pytest.mark.parametrize("k", [0, 4, 8])
def test_msckf_unit_test_scenario_through_cpp_facade(res, k):
    g = np.load(os.path.join(G, "msckf_unit_test.npz"))
    lay = o.layout(o.MULTI, k)
    for i in range(2):
        assert rel(res[f"msckf_k{k}_model_pred{i}_P"], g[f"k{k}_pred{i}_P"]) <= TOL
        assert np.abs(o.boxminus(lay, res[f"msckf_k{k}_model_pred{i}_mean"][:, 0], g[f"k{k}_pred{i}_mean"])).max() <= TOL
    assert rel(res[f"msckf_k{k}_model_upd_P"], g[f"k{k}_upd_P"]) <= TOL
    assert np.abs(o.boxminus(lay, res[f"msckf_k{k}_model_upd_mean"][:, 0], g[f"k{k}_upd_mean"])).max() <= TOL
    assert int(res[f"msckf_k{k}_model_outliers"][0, 0]) == int(g[f"k{k}_outliers"][0])
    assert int(res["msckf_status"][0, 0]) == 0


def test_functor_path_equals_registered_model_path(res):
    lay = o.layout(o.MULTI, 4)
    for key in ("pred0", "pred1", "upd"):
        assert rel(res[f"msckf_k4_functor_{key}_P"], res[f"msckf_k4_model_{key}_P"]) <= TOL
        assert np.abs(o.boxminus(lay, res[f"msckf_k4_functor_{key}_mean"][:, 0],
                                 res[f"msckf_k4_model_{key}_mean"][:, 0])).max() <= TOL


def test_usckf_unit_test_scenario_through_cpp_facade(res):
    g = np.load(os.path.join(G, "usckf_unit_test.npz"))
    lay = o.layout(o.AUGMENTED, 0, 3, 9)
    np.testing.assert_array_equal(res["usckf_ctor_P"], g["ctor_P"])
    np.testing.assert_array_equal(res["usckf_setm2_P"], g["setm2_P"])
    np.testing.assert_array_equal(res["usckf_setm2_mean"][:, 0], g["setm2_mean"])
    for i in range(2):
        assert rel(res[f"usckf_pred{i}_P"], g[f"pred{i}_P"]) <= TOL
        assert np.abs(o.boxminus(lay, res[f"usckf_pred{i}_mean"][:, 0], g[f"pred{i}_mean"])).max() <= TOL
    assert int(res["usckf_literal_update_status"][0, 0]) & 1     # SLK_ST_LLT_FAIL


def test_dead_reckon_model_through_cpp_facade(res):
    # slk::DeadReckonModel (src/core/DeadReckon.hpp:129-239 fused into predict) against the CPU oracle
    u = np.load(os.path.join(G, "dead_reckon.npz"))["u"][5]
    lay = o.layout(o.MULTI, 1)
    f = o.Msckf(1, o.identity_state(lay), 0.025 * np.eye(18))
    assert f.predict(o.pm_dead_reckon(u), 0.01 * np.eye(12)) == 0
    assert rel(res["dead_reckon_P"], f.P) <= TOL
    assert np.abs(o.boxminus(lay, res["dead_reckon_mean"][:, 0], f.mean)).max() <= TOL


def test_ekf_update_through_cpp_facade(res):
    # Msckf::update(z, h, H, R) EKF overload (Msckf.hpp:284-349) with a functor of the reference's h(mu_state, H) form
    k, N, m = 1, 18, 24
    i = np.arange(m)[:, None]
    j = np.arange(N)[None, :]
    H = np.sin(0.37 * i + 1.3 * j) + np.where((i % N) == j, 2.0, 0.0)
    H[:, 6:12] = 0.0
    zmean = np.cos(0.3 * np.arange(m))
    z = zmean + 0.1 * np.sin(1.0 * np.arange(m))
    z[6] += 30.0
    lay = o.layout(o.MULTI, k)
    f = o.Msckf(k, o.identity_state(lay), 0.025 * np.eye(N))
    st, no = f.update_ekf(z, zmean, H, 0.04 * np.eye(m))
    assert st == 0 and no == int(res["ekf_outliers"][0, 0]) and no >= 1
    assert rel(res["ekf_P"], f.P) <= TOL
    assert np.abs(o.boxminus(lay, res["ekf_mean"][:, 0], f.mean)).max() <= TOL


# ------------------------------------------------------------------ the reference's own model functions
@pytest.fixture(scope="module")
def ref():
    import __graft_entry__ as ge
    ge.build()
    import facade_build
    return facade_build.run(name="reference_models", std="c++14")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_reference_process_model_and_significance_tests_through_facade(ref, tag):
    """tests/cpp/reference_models.cpp: processModel of test/MsckfUnitTest.cpp:33-47 pasted unchanged, bound with
    std::bind like the reference's boost::bind (:200-205); variants a = predict(f, Q) + update(z, h, R), b = predict(f, QFn,
    Nk) + update(z, h, R, accept_mahalanobis_distance), c = ... + update with an arbitrary callable mt."""
    g = np.load(os.path.join(G, "msckf_unit_test.npz"))
    k = 4
    lay = o.layout(o.MULTI, k)
    for i in range(2):
        assert rel(ref[f"ref_msckf_{tag}_pred{i}_P"], g[f"k{k}_pred{i}_P"]) <= TOL
        assert np.abs(o.boxminus(lay, ref[f"ref_msckf_{tag}_pred{i}_mean"][:, 0], g[f"k{k}_pred{i}_mean"])).max() <= TOL
    assert ref[f"ref_msckf_{tag}_check_ok"][0, 0] == 1 and ref[f"ref_msckf_{tag}_check_cov"][0, 0] <= 1e-12
    assert ref[f"ref_msckf_{tag}_check_mean"][0, 0] <= 1e-12
    # the update (third feature a gross outlier) against the oracle from the predicted golden state
    r = o.Msckf(k, g[f"k{k}_pred1_mean"], g[f"k{k}_pred1_P"])
    feat = g[f"k{k}_feat"].copy()
    z = g[f"k{k}_z"].copy()
    z[4] += 3.0
    st, no = r.update(z, o.mm_feature_proj(feat), 0.01 * np.eye(8))
    assert st == 0 and no == 1 == int(ref[f"ref_msckf_{tag}_outliers"][0, 0])
    assert int(ref[f"ref_msckf_{tag}_status"][0, 0]) == 0
    assert rel(ref[f"ref_msckf_{tag}_upd_P"], r.P) <= TOL
    assert np.abs(o.boxminus(lay, ref[f"ref_msckf_{tag}_upd_mean"][:, 0], r.mean)).max() <= TOL
    # the arbitrary callable really ran on the host: once per block it looked at (4 blocks, one rejected -> 4 calls)
    assert int(ref[f"ref_msckf_{tag}_mt_calls"][0, 0]) == (4 if tag == "c" else 0)
    for key in ("upd_P", "upd_mean"):
        # (a caller-side significance test goes through the general step kernel, the built-in gate through the exact-shape
        # fast path: the same update in another operation order -- equal to rounding, not bit for bit)
        np.testing.assert_allclose(ref[f"ref_msckf_{tag}_{key}"], ref[f"ref_msckf_a_{key}"], rtol=0, atol=1e-13)


def test_nonconst_mustate_window_edit_through_facade(ref):
    """Msckf.hpp:381-395: muState().sensorsk.push_back(...) + setPk(...) == a fresh filter built from the edited state."""
    np.testing.assert_array_equal(ref["ref_window_P"], ref["ref_window_fresh_P"])
    np.testing.assert_array_equal(ref["ref_window_mean"], ref["ref_window_fresh_mean"])
    assert ref["ref_window_P"].shape == (24, 24)


def test_reference_usckf_models_through_facade(ref):
    """processModel / measurementModelVO of test/UsckfUnitTest.cpp:34-86 pasted unchanged (functor path) == registered
    models on the GPU == the oracle, on an SPD state of the unit-test shape; update(z, h, RFn, mt) with a callable."""
    nfk, nfkl, N = 3, 9, 48
    mean = np.zeros(39 + nfk + nfkl)
    for b in range(3):
        s = 13 * b
        mean[s:s + 3] = [0.5 + 0.1 * b, -0.3 + 0.05 * b, 1.0 - 0.2 * b]
        mean[s + 3:s + 7] = o.so3_exp(np.array([0.05 * (b + 1), -0.04 * b, 0.03 + 0.02 * b]))
        mean[s + 7:s + 10] = [0.3, -0.1 * b, 0.2]
        mean[s + 10:s + 13] = [0.01 * b, 0.02, -0.01]
    mean[39:42] = 2.0 + 0.5 * np.arange(3)
    mean[42:51] = 1.0 + 0.25 * np.arange(9)
    i, j = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    A = 0.007 * (((i * 7 + j * 13) % 11) - 5.0) / 5.0
    P = A @ A.T + 0.0025 * np.eye(N)
    D2R = np.pi / 180
    f = o.Usckf(nfk=nfk, nfkl=nfkl, mean=mean, P=P)
    pm = o.pm_const_velocity(np.array([1.0, 0.1, -0.2]), np.array([10 * D2R, -5 * D2R, 8 * D2R]), 0.01)
    z = np.array([2.05, 2.45, 3.1])
    for _ in range(2):
        assert f.predict(pm, 0.1 * 0.01 * np.eye(12)) == 0
        st, acc = f.update(z, o.mm_vo_relative(), 0.01 * np.eye(3))
        assert st == 0 and acc == 1
    lay = o.layout(o.AUGMENTED, 0, nfk, nfkl)
    for tag in ("model", "functor", "mt"):
        assert int(ref[f"ref_usckf_{tag}_status"][0, 0]) == 0
        assert rel(ref[f"ref_usckf_{tag}_P"], f.P) <= TOL, tag
        assert np.abs(o.boxminus(lay, ref[f"ref_usckf_{tag}_mean"][:, 0], f.mean)).max() <= TOL, tag
        assert ref[f"ref_usckf_{tag}_check_ok"][0, 0] == 1 and ref[f"ref_usckf_{tag}_check_cov"][0, 0] <= 1e-10
    assert int(ref["ref_usckf_mt_mt_calls"][0, 0]) == 2
    assert ref["ref_usckf_rejected_unchanged"][0, 0] == 1


def test_callers_own_matrix_types_through_facade():
    """tests/cpp/foreign_matrix.cpp: the caller's matrices / vectors are fixed-size types of its own (standing in for
    Eigen::Matrix<double, 12, 12> etc. of a Rock task), not slk::Matrix: constructor, predict's Q, update's z / R, the EKF's
    H, setPk / setPkSingleState / setMeasurement take them, getPk / getPkSingleState / PkSingleState convert into them.
    The same scenario on the facade's own types must print the same numbers, and the k = 8 update agrees with the oracle."""
    import __graft_entry__ as ge
    ge.build()
    import facade_build
    r = facade_build.run(name="foreign_matrix")
    keys = [k[len("foreign_"):] for k in r if k.startswith("foreign_")]
    assert len(keys) >= 19
    for k in keys:
        np.testing.assert_array_equal(r["foreign_" + k], r["own_" + k], err_msg=k)
    assert r["foreign_k8_upd_P"].shape == (60, 60) and r["foreign_k8_P12"].shape == (12, 12)
    np.testing.assert_array_equal(r["foreign_k8_P12"], r["foreign_k8_upd_P"][:12, :12])
    assert int(r["foreign_k8_status"][0, 0]) == 0 and int(r["foreign_k2_status"][0, 0]) == 0
    assert r["foreign_usckf_PkI"].shape == (12, 12)
    # EKF update behind a caller-side significance test: all blocks rejected -> nothing applied, 12 outliers; four blocks
    # rejected -> 16 < 18 rows left: skipped, SLK_ST_EKF_ROWS (16) reported
    P0 = 0.025 * np.eye(18)
    np.testing.assert_array_equal(r["foreign_ekf_none_P"], P0)
    assert int(r["foreign_ekf_none_outliers"][0, 0]) == 12 and int(r["foreign_ekf_none_status"][0, 0]) == 0
    np.testing.assert_array_equal(r["foreign_ekf_some_P"], P0)
    assert int(r["foreign_ekf_some_outliers"][0, 0]) == 4 and int(r["foreign_ekf_some_status"][0, 0]) == 16
    # k = 8 against the oracle: two predicts with the delta-pose model, one update with four features
    import scenarios as sc
    s = sc.msckf_unit_test(8)
    f = o.Msckf(8, s["mean"], s["P"])
    pm = o.pm_delta_pose(s["dpos"], s["dquat"], s["velocity"], s["angular_velocity"])
    for _ in range(2):
        assert f.predict(pm, s["Q"]) == 0
    feat = np.array([[0.5 * (j - 1.5), 0.3 * (1.5 - j), 5.0 + j, (j % 8) + 1] for j in range(4)])
    z = np.array([v for j in range(4) for v in (0.1 * (j - 1.0) * 0.5, 0.05 * (j + 0.5) * 0.5)])
    st, no = f.update(z, o.mm_feature_proj(feat), 0.01 * np.eye(8))
    assert st == 0 and no == int(r["foreign_k8_outliers"][0, 0])
    assert rel(r["foreign_k8_upd_P"], f.P) <= TOL
    assert np.abs(o.boxminus(o.layout(o.MULTI, 8), r["foreign_k8_upd_mean"][:, 0], f.mean)).max() <= TOL
