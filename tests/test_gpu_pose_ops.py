"""GPU: the batched pose-with-uncertainty ops through the C ABI against the golden vectors and
the oracle: slk_transform_compose (Transform.cpp:215-254), slk_dead_reckon_pose (DeadReckon.hpp:129-239, :306-330),
slk_adaptive_matrix (MeasurementModels.hpp:181-286) and its use as the R of Msckf::update."""
import os

import numpy as np
import pytest

from oracle import oracle as o
import scenarios as sc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-12


@pytest.fixture(scope="module")
def slk():
    import torch  # noqa: F401
    from slkpkg import slk as mod
    assert mod.device_count() > 0, "no MI355X visible"
    return mod


@pytest.fixture(scope="module")
def handle(slk):
    s = sc.synthetic_msckf(24, 0, m=2, seed=1)
    return slk.Msckf(s["mean"], s["P"])          # the ops run on a handle's batch / stream


def test_transform_compose_against_golden(slk, handle):
    g = np.load(os.path.join(G, "pose_ops.npz"))
    s = sc.synthetic_pose_ops()
    for name, c2, c1 in (("both", s["cov2"], s["cov1"]), ("first", s["cov2"], None), ("second", None, s["cov1"]), ("none", None, None)):
        t, c = handle.transform_compose(s["t2"], c2, s["t1"], c1)
        assert np.abs(t - g[f"compose_{name}_t"]).max() <= TOL, name
        assert np.abs(c - g[f"compose_{name}_cov"]).max() <= TOL, name
    # additive branch of DeadReckon::updatePose's Affine3d overload (DeadReckon.hpp:317-323)
    t, c = handle.transform_compose(s["t2"], s["cov2"], s["t1"], s["cov1"], additive=True)
    assert np.abs(t - g["compose_none_t"]).max() <= TOL and np.abs(c - (s["cov2"] + s["cov1"])).max() <= 1e-18
    # large rotations: the non-positive-trace branch of the matrix -> quaternion conversion, against the oracle
    rng = np.random.default_rng(9)
    t2 = np.concatenate([rng.normal(size=(24, 3)), sc.quat_exp(rng.normal(0, 2.0, (24, 3)))], axis=1)
    t1 = np.concatenate([rng.normal(size=(24, 3)), sc.quat_exp(rng.normal(0, 2.0, (24, 3)))], axis=1)
    t, c = handle.transform_compose(t2, s["cov2"], t1, s["cov1"])
    wmin = 1.0
    for b in range(24):
        to, co = o.transform_compose(t2[b], s["cov2"][b], t1[b], s["cov1"][b])
        assert np.abs(t[b] - to).max() <= 1e-11 and np.abs(c[b] - co).max() <= 1e-11 * max(1.0, np.abs(co).max())
        wmin = min(wmin, to[6])
    # composite quaternions with w < 0 are among them: q_to_r follows Eigen >= 3.3's AngleAxisd(q) there (angle 2 atan2(|vec|, |w|),
    # axis flipped), not the 2 acos(w) of Eigen 3.0 - 3.2 -- the reference pins no Eigen version and holds no fixture for
    # Transform: the chosen behaviour is pinned here, parity with the reference is not
    assert wmin < -0.1


def test_adaptive_attitude_cov_rank_deficient_history(slk):
    """m1 = 1: Uk is a single outer product (rank one, two zero singular values whose basis JacobiSVD leaves arbitrary,
    MeasurementModels.hpp:230-235): the GPU op and the oracle take the same basis (cyclic Jacobi from the identity); pinned
    against each other, parity with the reference unpinned."""
    s = sc.synthetic_pose_ops()
    B = s["B"]
    a = slk.AdaptiveAttitudeCov(B, 1, s["m2"], s["gamma"], s["r2count"])
    refs = [o.AdaptiveAttitudeCov(1, s["m2"], s["gamma"], s["r2count"]) for _ in range(B)]
    for k in range(6):
        R = a.matrix(s["xk"][k], s["Pk"], s["z"][k], s["H"][k], s["R"])
        for b in range(B):
            Rb = refs[b].matrix(s["xk"][k][b], s["Pk"][b] if np.ndim(s["Pk"]) == 3 else s["Pk"], s["z"][k][b], s["H"][k][b] if np.ndim(s["H"][k]) == 3 else s["H"][k],
                                s["R"][b] if np.ndim(s["R"]) == 3 else s["R"])
            assert np.abs(R[b] - Rb).max() <= 1e-12 * max(1.0, np.abs(Rb).max()), (k, b)


@pytest.mark.parametrize("tf", [0, 1])
def test_dead_reckon_pose_against_golden(slk, handle, tf):
    g = np.load(os.path.join(G, "pose_ops.npz"))
    s = sc.synthetic_pose_ops()
    post = np.concatenate([s["prev"], np.zeros((24, 24))], axis=1)
    po, de = handle.dead_reckon_pose(s["u"], s["velcov"], s["prev"], post, use_tf=bool(tf))
    assert np.abs(po - g[f"dr_pose_tf{tf}_post"]).max() <= TOL
    assert np.abs(de - g[f"dr_pose_tf{tf}_delta"]).max() <= TOL
    # the delta pose feeds predict(): the same numbers as the round-1 delta-pose op
    d13 = handle.dead_reckon(s["u"])
    assert np.abs(np.concatenate([de[:, :7], de[:, 25:]], axis=1) - d13).max() <= 1e-15
    # a NaN in the velocity covariance (per-filter covariances this time)
    vc = np.tile(s["velcov"], (24, 1, 1))
    vc[5, 0, 0] = np.nan
    po2, de2 = handle.dead_reckon_pose(s["u"], vc, s["prev"], post, use_tf=bool(tf))
    assert np.abs(de2[5, 7:25]).max() == 0.0 and np.abs(np.delete(de2, 5, 0) - np.delete(de, 5, 0)).max() <= 1e-15


def test_adaptive_attitude_cov_against_golden_and_as_update_noise(slk):
    g = np.load(os.path.join(G, "pose_ops.npz"))
    s = sc.synthetic_pose_ops()
    B = s["B"]
    a = slk.AdaptiveAttitudeCov(B, s["m1"], s["m2"], s["gamma"], s["r2count"])
    for k in range(s["steps"]):
        R = a.matrix(s["xk"][k], s["Pk"], s["z"][k], s["H"][k], s["R"])
        assert np.abs(R - g["adaptive_R"][k]).max() <= 1e-12, k
    # the adapted covariance as the per-filter R of update() (position fix of the current pose), against the oracle
    f = sc.synthetic_msckf(B, 2, m=2, seed=31)
    filt = slk.Msckf(f["mean"], f["P"])
    z3 = f["mean"][:, 0:3] + 0.03
    filt.update(z3, slk.MM_POSE_POSITION, np.array([0.0]), R, gate=0)
    assert (filt.status() == 0).all()
    lay = o.layout(o.MULTI, 2)
    Pg, Mg = filt.getPk(), filt.muState()
    for b in range(B):
        r = o.Msckf(2, f["mean"][b], f["P"][b])
        st, _ = r.update(z3[b], o.mm_pose_position(0), R[b], gate=False)
        assert st == 0
        assert np.abs(Pg[b] - r.P).max() / np.abs(r.P).max() <= 1e-9
        assert np.abs(o.boxminus(lay, Mg[b], r.mean)).max() <= 1e-9
