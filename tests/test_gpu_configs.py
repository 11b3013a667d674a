"""GPU parity tests for the BASELINE.json configurations the first round left unexercised, and for the
checkSigmaPoints self test on the HIP path:

  * cfg2 as benchmarked: B = 1024 filters, N = 12 (k = 0) with a 3-row position fix and N = 18 (k = 1) with one
    2-D feature -- sampled parity against the oracle plus size-independent properties;
  * cfg5: N = 198 (k = 31) and N = 204 (k = 32), covariance rebuild in fp64 (parity path, 1e-9), fp32 MFMA (1e-6)
    and bf16 operands / fp32 accumulation (2e-2, relative to max |P|), and the full B = 512 batch;
  * checkSigmaPoints (Msckf.hpp:819-839) evaluated by the library on the device;
  * rejected misuse: pose indices out of range (host- and device-resident parameters).
Run with `pytest -m gpu` on an MI355X.
"""
import numpy as np
import pytest

from oracle import oracle as o
import scenarios as sc

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def slk():
    import torch  # noqa: F401
    from slkpkg import slk as mod
    assert mod.device_count() > 0, "no MI355X visible"
    return mod


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def mean_err(lay, a, b):
    return float(np.abs(o.boxminus(lay, a, b)).max())


# ------------------------------------------------------------------ cfg2: N = 12 / 18, B = 1024
def test_cfg2_k0_position_fix_full_batch(slk):
    """bench.py --clones 0 --meas 3: delta-pose predict + 3-row position measurement of pose 0, ungated."""
    B, k = 1024, 0
    s = sc.synthetic_msckf(B, k, m=2, seed=0x5EED0000)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    R = 0.01 * np.eye(3)
    z3 = s["mean"][:, 0:3] + 0.05
    f = slk.Msckf(s["mean"], s["P"])
    steps = 2
    for _ in range(steps):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], z3, slk.MM_POSE_POSITION, np.array([0.0]), R, gate=0)
    assert (f.status() == 0).all() and (f.outliers() == 0).all()
    P, M = f.getPk(), f.muState()
    assert np.isfinite(P).all() and np.isfinite(M).all()
    np.testing.assert_array_equal(P, np.transpose(P, (0, 2, 1)))
    assert np.linalg.eigvalsh(P).min() > 0
    np.testing.assert_allclose(np.linalg.norm(M[:, 3:7], axis=-1), 1.0, atol=1e-12)
    for b in np.r_[0:12, B - 12:B]:
        r = o.Msckf(k, s["mean"][b], s["P"][b])
        for _ in range(steps):
            assert r.predict(o.pm_delta_pose(s["u"][b, 0:3], s["u"][b, 3:7], s["u"][b, 7:10], s["u"][b, 10:13]), s["Q"]) == 0
            st, no = r.update(z3[b], o.mm_pose_position(0), R, gate=False)
            assert st == 0 and no == 0
        assert rel(P[b], r.P) <= TOL, b
        assert mean_err(lay, M[b], r.mean) <= TOL, b


def test_cfg2_k1_one_feature_full_batch(slk):
    """bench.py --clones 1 --meas 2: N = 18, one 2-D feature seen from the clone, chi-square gate on."""
    B, k, m = 1024, 1, 2
    s = sc.synthetic_msckf(B, k, m=m, seed=0x5EED0000)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    f = slk.Msckf(s["mean"], s["P"])
    steps = 2
    tot = np.zeros(B, dtype=np.int64)
    for _ in range(steps):
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        tot += f.outliers()
    st = f.status()
    assert (st & ~slk.ST_ALL_REJECTED == 0).all()
    P, M = f.getPk(), f.muState()
    assert np.isfinite(P).all() and np.isfinite(M).all()
    assert np.linalg.eigvalsh(0.5 * (P + np.transpose(P, (0, 2, 1)))).min() > 0
    idx = np.r_[0:16, B - 16:B]
    mean, Pc = s["mean"][idx].copy(), s["P"][idx].copy()
    stc, out = o.msckf_step_batch(k, m, steps, mean, Pc, np.ascontiguousarray(s["u"][idx]),
                                  np.ascontiguousarray(s["feat"][idx]), np.ascontiguousarray(s["z"][idx]), s["Q"], s["R"])
    assert stc == 0
    np.testing.assert_array_equal(tot[idx], out)
    for j, b in enumerate(idx):
        assert rel(P[b], Pc[j].reshape(N, N).T) <= TOL, b
        assert mean_err(lay, M[b], mean[j]) <= TOL, b


# ------------------------------------------------------------------ cfg5: N = 198 / 204, rebuild precisions
@pytest.mark.parametrize("k", [31, 32])
def test_cfg5_rebuild_precisions_against_oracle(slk, k):
    B, m = 4, 8
    s = sc.synthetic_msckf(B, k, m=m, seed=500 + k)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    mean, P = s["mean"].copy(), s["P"].copy()
    st, out = o.msckf_step_batch(k, m, 1, mean, P, s["u"], s["feat"], s["z"], s["Q"], s["R"])
    assert st == 0
    # bounds on max |P - P_oracle| / max |P_oracle|: fp64 = parity path; fp32 MFMA ~2.5e-7; bf16 operands ~1e-2
    for mode, bound in ((0, TOL), (1, 1e-6), (2, 2e-2)):
        f = slk.Msckf(s["mean"], s["P"])
        f.set_rebuild_precision(mode)
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
        assert (f.status() & ~slk.ST_ALL_REJECTED == 0).all()
        np.testing.assert_array_equal(f.outliers(), out)
        Pg, Mg = f.getPk(), f.muState()
        worst = max(rel(Pg[b], P[b].reshape(N, N).T) for b in range(B))
        assert worst <= bound, (mode, worst)
        if mode:                       # the reduced modes must really be reduced (not silently the fp64 path)
            assert worst > 1e-11, (mode, worst)
        for b in range(B):             # the mean never goes through the reduced arithmetic
            assert mean_err(lay, Mg[b], mean[b]) <= TOL, (mode, b)


def test_cfg5_full_batch_properties_and_sampled_parity(slk):
    B, k, m = 512, 31, 8
    s = sc.synthetic_msckf(B, k, m=m, seed=0x5EED0000)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    f = slk.Msckf(s["mean"], s["P"])
    f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    tot = f.outliers()
    st = f.status()
    assert (st & ~slk.ST_ALL_REJECTED == 0).all()
    P, M = f.getPk(), f.muState()
    upd = st == 0
    assert np.isfinite(P).all() and np.isfinite(M).all()
    np.testing.assert_array_equal(P[upd], np.transpose(P[upd], (0, 2, 1)))
    assert np.linalg.eigvalsh(P[::8]).min() > 0
    q = np.concatenate([M[:, 3:7][:, None, :]] + [M[:, 13 + 7 * c + 3:13 + 7 * c + 7][:, None, :] for c in range(k)], axis=1)
    np.testing.assert_allclose(np.linalg.norm(q, axis=-1), 1.0, atol=1e-12)
    idx = np.r_[0:4, B - 4:B]
    mean, Pc = s["mean"][idx].copy(), s["P"][idx].copy()
    stc, out = o.msckf_step_batch(k, m, 1, mean, Pc, np.ascontiguousarray(s["u"][idx]),
                                  np.ascontiguousarray(s["feat"][idx]), np.ascontiguousarray(s["z"][idx]), s["Q"], s["R"])
    assert stc == 0
    np.testing.assert_array_equal(tot[idx], out)
    for j, b in enumerate(idx):
        assert rel(P[b], Pc[j].reshape(N, N).T) <= TOL
        assert mean_err(lay, M[b], mean[j]) <= TOL


# ------------------------------------------------------------------ checkSigmaPoints on the HIP path
@pytest.mark.parametrize("k", [0, 2, 8, 13, 31])
def test_check_sigma_points_on_the_device(slk, k):
    """Msckf.hpp:819-839: cov(sigma points of (mu, Pk)) == Pk and mean == mu.  Two routes: the library's own
    slk_check_sigma_points (all on the GPU) and the emitted sigma points (slk_update_sigma_points) folded on the host."""
    B = 6
    s = sc.synthetic_msckf(B, k, m=2, seed=700 + k)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    f = slk.Msckf(s["mean"], s["P"])
    ce, me = f.checkSigmaPoints()
    assert (f.status() == 0).all()
    scale = np.abs(s["P"]).max()
    assert ce.max() <= 1e-12 * max(1.0, scale / 1e-2) and me.max() <= 1e-12, (ce.max(), me.max())
    # the filter is untouched
    np.testing.assert_array_equal(f.getPk(), s["P"])
    np.testing.assert_array_equal(f.muState(), s["mean"])
    if k <= 8:
        X = f.update_sigma_points()
        S = 2 * N + 1
        assert X.shape == (B, S, s["Nq"])
        for b in range(B):
            D = np.array([o.boxminus(lay, X[b, i], s["mean"][b]) for i in range(S)])
            assert np.abs(D.sum(axis=0) / S).max() <= 1e-12               # the mean of the sigma points is mu
            assert np.abs(0.5 * D.T @ D - s["P"][b]).max() <= 1e-12        # 1/2 sum d d^T == Pk
    # the oracle's statement of the same invariant agrees
    r = o.Msckf(k, s["mean"][0], s["P"][0])
    st, a, b2 = r.check_sigma_points()
    assert st == 0 and a <= 1e-12 and b2 <= 1e-12


# ------------------------------------------------------------------ misuse: pose indices
def test_pose_index_out_of_range_is_rejected(slk):
    import torch
    s = sc.synthetic_msckf(4, 2, m=4, seed=12)
    f = slk.Msckf(s["mean"], s["P"])
    for bad in (-1.0, 3.0, float("nan"), 1e9):
        feat = s["feat"].copy()
        feat[2, 1, 3] = bad
        with pytest.raises(slk.SlkError):                       # host parameters: refused before any launch
            f.update(s["z"], slk.MM_FEATURE_PROJ, feat, s["R"])
        with pytest.raises(slk.SlkError):
            f.update(np.zeros((4, 3)), slk.MM_POSE_POSITION, np.array([bad]), 0.01 * np.eye(3), gate=0)
    np.testing.assert_array_equal(f.getPk(), s["P"])
    # device-resident parameters: the kernel reports the filter and skips its update, the others are updated
    feat = s["feat"].copy()
    feat[2, 1, 3] = 7.0
    dev = torch.device("cuda", 0)
    d = {n: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for n, v in
         (("feat", feat.reshape(4, -1)), ("z", s["z"]), ("R", s["R"]))}
    f.update(d["z"], slk.MM_FEATURE_PROJ, d["feat"], d["R"])
    st = f.status()
    assert st[2] == slk.ST_BAD_INDEX and (np.delete(st, 2) & ~slk.ST_ALL_REJECTED == 0).all()
    np.testing.assert_array_equal(f.getPk()[2], s["P"][2])
    np.testing.assert_array_equal(f.muState()[2], s["mean"][2])
    g = slk.Msckf(s["mean"], s["P"])
    g.update(s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    for b in (0, 1, 3):
        np.testing.assert_array_equal(f.getPk()[b], g.getPk()[b])


def test_shared_measurement_row_is_broadcast(slk):
    """A 1-D z is one measurement for every filter of the batch (the C ABI always reads [B][m])."""
    s = sc.synthetic_msckf(3, 2, m=4, seed=13)
    s["mean"][:] = s["mean"][0]
    s["P"][:] = s["P"][0]
    a = slk.Msckf(s["mean"], s["P"])
    b = slk.Msckf(s["mean"], s["P"])
    a.update(s["z"][0], slk.MM_FEATURE_PROJ, s["feat"][0].reshape(-1), s["R"])
    b.update(np.tile(s["z"][0], (3, 1)), slk.MM_FEATURE_PROJ, np.tile(s["feat"][0], (3, 1, 1)), s["R"])
    np.testing.assert_array_equal(a.getPk(), b.getPk())
    np.testing.assert_array_equal(a.getPk()[0], a.getPk()[2])


def test_two_devices_in_one_process(slk):
    """A process may own handles on several GPUs (slk_config.device): both must match the oracle."""
    if slk.device_count() < 2:
        pytest.skip("one GPU visible")
    k, m = 8, 8
    s = sc.synthetic_msckf(8, k, m=m, seed=99)
    lay = o.layout(o.MULTI, k)
    N = s["N"]
    fs = [slk.Msckf(s["mean"], s["P"], device=d) for d in (0, 1)]
    for f in fs:
        f.step(slk.PM_DELTA_POSE, s["u"], s["Q"], s["z"], slk.MM_FEATURE_PROJ, s["feat"], s["R"])
    mean, P = s["mean"].copy(), s["P"].copy()
    st, out = o.msckf_step_batch(k, m, 1, mean, P, s["u"], s["feat"], s["z"], s["Q"], s["R"])
    for f in fs:
        assert (f.status() & ~slk.ST_ALL_REJECTED == 0).all()
        for b in range(8):
            assert rel(f.getPk()[b], P[b].reshape(N, N).T) <= TOL and mean_err(lay, f.muState()[b], mean[b]) <= TOL
