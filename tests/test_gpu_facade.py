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
