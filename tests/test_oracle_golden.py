"""The CPU oracle against the committed golden fixtures (tests/golden/*.npz, written by
tests/golden/make_golden.py) and against the independent numpy/scipy implementation (G6).
Golden vectors are oracle-generated: the reference holds none ("parity unpinned")."""
import os

import numpy as np
import pytest

from oracle import np_check as npc
from oracle import oracle as o
import scenarios as sc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-12


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def test_usckf_unit_test_golden():
    g = np.load(os.path.join(G, "usckf_unit_test.npz"))
    u = sc.usckf_unit_test()
    f = o.Usckf(state13=u["state_single"], P0_12=u["P0_single"])
    assert rel(f.P, g["ctor_P"]) <= TOL
    for i, (mode, z, R) in enumerate(u["set_measurements"]):
        f.set_measurement(mode, z, R)
        assert rel(f.P, g[f"setm{i}_P"]) <= TOL
        np.testing.assert_allclose(f.mean, g[f"setm{i}_mean"], rtol=TOL, atol=TOL)
    pm = o.pm_const_velocity(u["velocity"], u["angular_velocity"], u["dt"])
    for i in range(u["n_predict"]):
        assert f.predict(pm, u["Q"]) == 0
        assert rel(f.P, g[f"pred{i}_P"]) <= TOL
        assert np.abs(o.boxminus(f.lay, f.mean, g[f"pred{i}_mean"])).max() <= TOL
    assert int(g["literal_update_status"][0]) & o.LLT_FAIL


@pytest.mark.parametrize("k", [0, 1, 4, 8, 31])
def test_msckf_unit_test_golden(k):
    g = np.load(os.path.join(G, "msckf_unit_test.npz"))
    t = sc.msckf_unit_test(k)
    f = o.Msckf(k, t["mean"], t["P"])
    pm = o.pm_delta_pose(t["dpos"], t["dquat"], t["velocity"], t["angular_velocity"])
    for i in range(t["n_predict"]):
        assert f.predict(pm, t["Q"]) == 0
        assert rel(f.P, g[f"k{k}_pred{i}_P"]) <= TOL
        assert np.abs(o.boxminus(f.lay, f.mean, g[f"k{k}_pred{i}_mean"])).max() <= TOL
    R = 0.01 * np.eye(g[f"k{k}_z"].size)
    st, no = f.update(g[f"k{k}_z"], o.mm_feature_proj(g[f"k{k}_feat"]), R)
    assert st == 0 and no == int(g[f"k{k}_outliers"][0])
    assert rel(f.P, g[f"k{k}_upd_P"]) <= TOL
    assert np.abs(o.boxminus(f.lay, f.mean, g[f"k{k}_upd_mean"])).max() <= TOL


def test_msckf_batch_golden_and_np_crosscheck():
    g = np.load(os.path.join(G, "msckf_batch.npz"))
    s = sc.synthetic_msckf(8, 8)
    # through the batch driver (the cpu_baseline code path)
    mean, P = s["mean"].copy(), s["P"].copy()
    st, out = o.msckf_step_batch(8, 8, 3, mean, P, s["u"], s["feat"], g["z"], s["Q"], s["R"])
    assert st == 0
    np.testing.assert_array_equal(out, g["outliers"])
    lay = o.layout(o.MULTI, 8)
    for b in range(8):
        Pb = P[b].reshape(60, 60).T                 # batch driver keeps column-major per filter
        assert rel(Pb, g["P"][b]) <= TOL
        assert np.abs(o.boxminus(lay, mean[b], g["mean"][b])).max() <= TOL
    # independent numpy/scipy implementation on one filter with outliers (b = 2)
    b = 2
    gnp = npc.Msckf(8, s["mean"][b], s["P"][b])
    u = s["u"][b]
    tot = 0
    for _ in range(3):
        gnp.predict(lambda x: npc.pm_delta_pose(x, u[0:3], u[3:7], u[7:10], u[10:13]), s["Q"])
        tot += gnp.update(g["z"][b], lambda X: npc.mm_feature_proj(X, s["feat"][b]), s["R"])
    assert tot == int(g["outliers"][b]) and tot > 0
    assert rel(gnp.P, g["P"][b]) <= TOL
    assert np.abs(o.boxminus(lay, gnp.mean, g["mean"][b])).max() <= TOL


def test_usckf_spd_golden_and_np_crosscheck():
    g = np.load(os.path.join(G, "usckf_spd.npz"))
    s = sc.synthetic_usckf(4)
    for b in range(4):
        f = o.Usckf(nfk=3, nfkl=9, mean=s["mean"][b], P=s["P"][b])
        gnp = npc.Usckf(3, 9, s["mean"][b], s["P"][b])
        u = s["u"][b]
        pm = o.pm_const_velocity(u[0:3], u[3:6], u[6])
        for _ in range(2):
            assert f.predict(pm, s["Q"]) == 0
            st, acc = f.update(s["z"][b], o.mm_vo_relative(), s["R"])
            assert st == 0 and acc == 1
            gnp.predict(lambda x: npc.pm_const_velocity(x, u[0:3], u[3:6], u[6]), s["Q"])
            gnp.update(s["z"][b], lambda X: npc.mm_vo_relative(X, 3), s["R"])
        assert rel(f.P, g["P"][b]) <= TOL and rel(gnp.P, g["P"][b]) <= TOL
        assert np.abs(o.boxminus(f.lay, f.mean, g["mean"][b])).max() <= TOL
        assert np.abs(o.boxminus(f.lay, gnp.mean, g["mean"][b])).max() <= TOL


def test_usckf_batch_driver_equals_the_golden_objects():
    # slko_usckf_step_batch (the CPU-baseline leg of `bench.py --filter usckf`) is the same two calls per step
    g = np.load(os.path.join(G, "usckf_spd.npz"))
    s = sc.synthetic_usckf(4)
    mean, P = s["mean"].copy(), np.ascontiguousarray(np.transpose(s["P"], (0, 2, 1))).reshape(4, -1)
    assert o.usckf_step_batch(3, 9, 2, mean, P, s["u"], s["z"], s["Q"], s["R"]) == 0
    lay = o.layout(o.AUGMENTED, 0, 3, 9)
    for b in range(4):
        assert rel(P[b].reshape(48, 48).T, g["P"][b]) <= TOL
        assert np.abs(o.boxminus(lay, mean[b], g["mean"][b])).max() <= TOL


def test_dead_reckon_golden_and_np_crosscheck():
    # DeadReckon::updatePose delta pose (src/core/DeadReckon.hpp:129-239, updateAttitude :246-286) and two
    # Msckf predicts driven by it (SLK_PM_DEAD_RECKON)
    g = np.load(os.path.join(G, "dead_reckon.npz"))
    u = g["u"]
    d = o.dead_reckon_delta(u)
    assert np.abs(d - g["delta"]).max() <= 1e-15
    e = np.array([npc.dead_reckon_delta(r) for r in u])        # the reference's full 4x4 expression
    assert np.abs(e - g["delta"]).max() <= 1e-15
    assert np.allclose(d[0], [0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0])            # standing still
    assert np.allclose(d[1, 3:7], [0, 0, 0, 1]) and np.allclose(d[1, 0:3], 0.5 * u[1, 0] * (u[1, 1:4] + u[1, 7:10]))
    assert np.allclose(np.linalg.norm(d[:, 3:7], axis=1), 1.0, atol=1e-15)
    # constant angular velocity about one axis: third-order integration vs the exact rotation
    dt, w = 0.01, np.array([0.0, 0.0, 0.7])
    q = o.dead_reckon_delta(np.concatenate([[dt], np.zeros(3), w, np.zeros(3), w]))[3:7]
    assert abs(2.0 * np.arctan2(q[2], q[3]) - dt * 0.7) < 1e-9
    s = sc.synthetic_msckf(8, 2, m=2, seed=77)
    for b in range(8):
        f = o.Msckf(2, s["mean"][b], s["P"][b])
        for step in range(2):
            assert f.predict(o.pm_dead_reckon(u[8 * step + b]), s["Q"]) == 0
        assert float(np.abs(o.boxminus(f.lay, f.mean, g["mean"][b])).max()) <= 1e-13
        assert rel(f.P, g["P"][b]) <= 1e-13


@pytest.mark.parametrize("k,m", [(1, 24), (4, 48), (8, 72), (8, 128)])
def test_msckf_ekf_update_golden_and_np_crosscheck(k, m):
    # Msckf EKF update (Msckf.hpp:284-349; removeOutliers :756-789, reduceDimension :791-816)
    g = np.load(os.path.join(G, "msckf_ekf.npz"))
    e = sc.synthetic_ekf(3, k, m, seed=0xEC0F + k + m)
    lay = o.layout(o.MULTI, k)
    for b in range(3):
        f = o.Msckf(k, e["mean"][b], e["P"][b])
        st, no = f.update_ekf(e["z"][b], e["zmean"][b], e["H"][b], e["R"][b])
        assert st == 0 and no == int(g[f"k{k}_m{m}_outliers"][b])
        assert float(np.abs(o.boxminus(lay, f.mean, g[f"k{k}_m{m}_mean"][b])).max()) <= 1e-12
        assert rel(f.P, g[f"k{k}_m{m}_P"][b]) <= 1e-12
        h = npc.Msckf(k, e["mean"][b], e["P"][b])
        no2, flag = npc.msckf_update_ekf(h, e["z"][b], e["zmean"][b], e["H"][b], e["R"][b])
        assert flag is None and no2 == no
        assert float(np.abs(o.boxminus(lay, f.mean, h.mean)).max()) <= 1e-11 and rel(f.P, h.P) <= 1e-11


def test_msckf_ekf_update_known_answers():
    # gate off, H of full column rank and R = s^2 I: the compression is lossless, the result is the textbook EKF update
    k, m = 2, 60
    rng = np.random.default_rng(77)
    e = sc.synthetic_ekf(1, k, m, seed=321, outliers=False)
    N = e["N"]
    H = rng.normal(0, 1, (m, N))
    R = 0.04 * np.eye(m)
    P, mean = e["P"][0], e["mean"][0]
    z, zmean = e["z"][0], e["zmean"][0]
    f = o.Msckf(k, mean, P)
    st, no = f.update_ekf(z, zmean, H, R, gate=False)
    assert st == 0 and no == 0
    S = H @ P @ H.T + R
    K = P @ H.T @ np.linalg.inv(S)
    lay = o.layout(o.MULTI, k)
    assert rel(f.P, P - K @ S @ K.T) <= 1e-10
    assert float(np.abs(o.boxminus(lay, f.mean, o.boxplus(lay, mean, K @ (z - zmean)))).max()) <= 1e-10
    # fewer surviving rows than state dimensions: the reference would index R.block(0,0,N,N) out of range (:806)
    f = o.Msckf(k, mean, P)
    st, no = f.update_ekf(z[:N - 4], zmean[:N - 4], H[:N - 4], R[:N - 4, :N - 4], gate=False)
    assert st == 16 and rel(f.P, P) == 0.0
