"""Generate the golden fixtures under tests/golden/ (run from the repo root:
`python tests/golden/make_golden.py`).

The reference holds NO golden vectors for predict()/update() and cannot be built here
(SURVEY.md 8c), so these vectors are produced by the CPU oracle (oracle/slk_oracle.c) and
are only written after the independent numpy/scipy implementation (oracle/np_check.py)
agrees to <= 1e-12 -- "parity unpinned" with respect to the reference itself.
Inputs come from tests/scenarios.py (reference unit-test scenarios + seeded synthetic batches).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import np_check as npc  # noqa: E402
from oracle import oracle as o      # noqa: E402
import scenarios as sc              # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-12


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def check(lay, f_mean, f_P, g_mean, g_P, what):
    em = float(np.abs(o.boxminus(lay, f_mean, g_mean)).max())
    ep = rel(f_P, g_P)
    assert em <= TOL and ep <= TOL, (what, em, ep)


def usckf_unit_test():
    u = sc.usckf_unit_test()
    f = o.Usckf(state13=u["state_single"], P0_12=u["P0_single"])
    out = {"ctor_mean": f.mean, "ctor_P": f.P}
    for i, (mode, z, R) in enumerate(u["set_measurements"]):
        f.set_measurement(mode, z, R)
        out[f"setm{i}_mean"], out[f"setm{i}_P"] = f.mean, f.P
    g = npc.Usckf(f.lay.nfk, f.lay.nfkl, f.mean, f.P)
    pm = o.pm_const_velocity(u["velocity"], u["angular_velocity"], u["dt"])
    for i in range(u["n_predict"]):
        st = f.predict(pm, u["Q"])
        assert st == 0
        g.predict(lambda x: npc.pm_const_velocity(x, u["velocity"], u["angular_velocity"], u["dt"]), u["Q"])
        check(f.lay, f.mean, f.P, g.mean, g.P, f"usckf predict {i}")
        out[f"pred{i}_mean"], out[f"pred{i}_P"] = f.mean, f.P
    # literal update(): the 48x48 LLT hits a non-positive pivot
    st, _ = f.update(u["z"], o.mm_vo_relative(), u["R"])
    out["literal_update_status"] = np.array([st])
    np.savez(os.path.join(OUT, "usckf_unit_test.npz"), **out)


def usckf_spd():
    s = sc.synthetic_usckf(4)
    means, Ps = [], []
    for b in range(s["B"]):
        f = o.Usckf(nfk=s["nfk"], nfkl=s["nfkl"], mean=s["mean"][b], P=s["P"][b])
        g = npc.Usckf(s["nfk"], s["nfkl"], s["mean"][b], s["P"][b])
        u = s["u"][b]
        pm = o.pm_const_velocity(u[0:3], u[3:6], u[6])
        for step in range(2):
            assert f.predict(pm, s["Q"]) == 0
            g.predict(lambda x: npc.pm_const_velocity(x, u[0:3], u[3:6], u[6]), s["Q"])
            st, acc = f.update(s["z"][b], o.mm_vo_relative(), s["R"])
            assert st == 0 and acc == 1
            g.update(s["z"][b], lambda X: npc.mm_vo_relative(X, s["nfk"]), s["R"])
            check(f.lay, f.mean, f.P, g.mean, g.P, f"usckf spd {b} {step}")
        means.append(f.mean)
        Ps.append(f.P)
    np.savez(os.path.join(OUT, "usckf_spd.npz"), mean=np.array(means), P=np.array(Ps))


def msckf_unit_test():
    out = {}
    for k in (0, 1, 4, 8, 31):
        t = sc.msckf_unit_test(k)
        f = o.Msckf(k, t["mean"], t["P"])
        g = npc.Msckf(k, t["mean"], t["P"])
        pm = o.pm_delta_pose(t["dpos"], t["dquat"], t["velocity"], t["angular_velocity"])
        for i in range(t["n_predict"]):
            assert f.predict(pm, t["Q"]) == 0
            g.predict(lambda x: npc.pm_delta_pose(x, t["dpos"], t["dquat"], t["velocity"], t["angular_velocity"]), t["Q"])
            check(f.lay, f.mean, f.P, g.mean, g.P, f"msckf ut predict k={k} {i}")
            out[f"k{k}_pred{i}_mean"], out[f"k{k}_pred{i}_P"] = f.mean, f.P
        out[f"k{k}_Fk"] = f.Fk
        # update with the registered feature-projection model (the reference test stops before update)
        nf = 4 if k > 0 else 1
        feat = np.array([[0.5 * (j - 1.5), 0.3 * (1.5 - j), 5.0 + j, (j % k) + 1 if k else 0] for j in range(nf)], float)
        z = np.array([[0.1 * (j - 1.0), 0.05 * (j + 0.5)] for j in range(nf)], float).reshape(-1) * 0.5
        R = 0.01 * np.eye(2 * nf)
        st, no = f.update(z, o.mm_feature_proj(feat), R)
        assert st == 0
        no2 = g.update(z, lambda X: npc.mm_feature_proj(X, feat), R)
        assert no == no2
        check(f.lay, f.mean, f.P, g.mean, g.P, f"msckf ut update k={k}")
        out[f"k{k}_feat"], out[f"k{k}_z"] = feat, z
        out[f"k{k}_upd_mean"], out[f"k{k}_upd_P"], out[f"k{k}_outliers"] = f.mean, f.P, np.array([no])
    np.savez(os.path.join(OUT, "msckf_unit_test.npz"), **out)


def msckf_batch():
    """Seeded synthetic batch (k=8, m=8) incl. forced outliers, 3 steps."""
    s = sc.synthetic_msckf(8, 8)
    z = s["z"].copy()
    z[1, 2:4] += 3.0          # feature 1 of filter 1: gross outlier
    z[2, 0:2] += 3.0          # feature 0 of filter 2
    z[2, 6:8] -= 3.0          # feature 3 (last block) of filter 2
    z[3, :] += 3.0            # every feature of filter 3 rejected -> update skipped
    means, Ps, outs = [], [], []
    for b in range(s["B"]):
        f = o.Msckf(8, s["mean"][b], s["P"][b])
        g = npc.Msckf(8, s["mean"][b], s["P"][b])
        u = s["u"][b]
        pm = o.pm_delta_pose(u[0:3], u[3:7], u[7:10], u[10:13])
        tot = 0
        for step in range(3):
            assert f.predict(pm, s["Q"]) == 0
            g.predict(lambda x: npc.pm_delta_pose(x, u[0:3], u[3:7], u[7:10], u[10:13]), s["Q"])
            st, no = f.update(z[b], o.mm_feature_proj(s["feat"][b]), s["R"])
            assert st == 0
            no2 = g.update(z[b], lambda X: npc.mm_feature_proj(X, s["feat"][b]), s["R"])
            assert no == no2, (b, step, no, no2)
            tot += no
            check(f.lay, f.mean, f.P, g.mean, g.P, f"msckf batch {b} {step}")
        means.append(f.mean)
        Ps.append(f.P)
        outs.append(tot)
    assert outs[1] > 0 and outs[2] > 0 and outs[3] >= 4, outs
    np.savez(os.path.join(OUT, "msckf_batch.npz"), z=z, mean=np.array(means), P=np.array(Ps), outliers=np.array(outs))


def dead_reckon():
    """DeadReckon::updatePose delta poses (src/core/DeadReckon.hpp:129-239, :246-286): oracle, cross-checked against the
    numpy restatement that evaluates the reference's full 4x4 expression; plus two Msckf predicts driven by them."""
    rng = np.random.default_rng(0x5EED0DE)
    n = 64
    u = np.concatenate([rng.uniform(0.005, 0.1, (n, 1)), rng.normal(0, 1.0, (n, 3)), rng.normal(0, 0.5, (n, 3)),
                        rng.normal(0, 1.0, (n, 3)), rng.normal(0, 0.5, (n, 3))], axis=1)
    u[0, 1:] = 0.0                      # standing still: identity delta
    u[1, 4:7] = u[1, 10:13] = 0.0       # no rotation
    d = o.dead_reckon_delta(u)
    e = np.array([npc.dead_reckon_delta(r) for r in u])
    assert np.abs(d - e).max() <= 1e-15, np.abs(d - e).max()
    assert np.allclose(d[0], [0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
    s = sc.synthetic_msckf(8, 2, m=2, seed=77)
    means, Ps = [], []
    for b in range(8):
        f = o.Msckf(2, s["mean"][b], s["P"][b])
        g = npc.Msckf(2, s["mean"][b], s["P"][b])
        for step in range(2):
            assert f.predict(o.pm_dead_reckon(u[8 * step + b]), s["Q"]) == 0
            g.predict(lambda x: npc.pm_dead_reckon(x, u[8 * step + b]), s["Q"])
            check(f.lay, f.mean, f.P, g.mean, g.P, f"dead reckon predict {b} {step}")
        means.append(f.mean)
        Ps.append(f.P)
    # inputs of the predicts: scenarios.synthetic_msckf(8, 2, m=2, seed=77), regenerated by the tests
    np.savez(os.path.join(OUT, "dead_reckon.npz"), u=u, delta=d, mean=np.array(means), P=np.array(Ps))


def msckf_ekf():
    """Msckf EKF update (Msckf.hpp:284-349): oracle vs the numpy/LAPACK restatement, with outliers, rank-deficient
    Jacobians (zero velocity columns) and a full (non-isotropic) R."""
    out = {}
    for (k, m) in ((1, 24), (4, 48), (8, 72), (8, 128)):
        e = sc.synthetic_ekf(3, k, m, seed=0xEC0F + k + m)
        means, Ps, outs = [], [], []
        for b in range(3):
            f = o.Msckf(k, e["mean"][b], e["P"][b])
            g = npc.Msckf(k, e["mean"][b], e["P"][b])
            st, no = f.update_ekf(e["z"][b], e["zmean"][b], e["H"][b], e["R"][b])
            no2, flag = npc.msckf_update_ekf(g, e["z"][b], e["zmean"][b], e["H"][b], e["R"][b])
            assert st == 0 and flag is None and no == no2, (k, m, b, st, no, no2, flag)
            check(f.lay, f.mean, f.P, g.mean, g.P, f"ekf {k} {m} {b}")
            means.append(f.mean); Ps.append(f.P); outs.append(no)
        assert sum(outs) > 0
        out[f"k{k}_m{m}_mean"] = np.array(means)
        out[f"k{k}_m{m}_P"] = np.array(Ps)
        out[f"k{k}_m{m}_outliers"] = np.array(outs)
    np.savez(os.path.join(OUT, "msckf_ekf.npz"), **out)


def pose_ops():
    """TransformWithUncertainty::operator* (Transform.cpp:215-254), both overloads of
    DeadReckon::updatePose (DeadReckon.hpp:129-239, :306-330) and AdaptiveAttitudeCov::matrix
    (MeasurementModels.hpp:181-286): oracle output, written once the numpy twin agrees (quaternions up to sign)."""
    s = sc.synthetic_pose_ops()
    B = s["B"]

    def same_transform(a, b):
        return max(np.abs(a[:3] - b[:3]).max(), min(np.abs(a[3:] - b[3:]).max(), np.abs(a[3:] + b[3:]).max()))
    out = {}
    for name, use2, use1 in (("both", True, True), ("first", True, False), ("second", False, True), ("none", False, False)):
        T, Cv = [], []
        for b in range(B):
            c2 = s["cov2"][b] if use2 else None
            c1 = s["cov1"][b] if use1 else None
            t, c = o.transform_compose(s["t2"][b], c2, s["t1"][b], c1)
            tn, cn = npc.transform_compose(s["t2"][b], c2, s["t1"][b], c1)
            assert same_transform(t, tn) <= TOL and np.abs(c - cn).max() <= TOL * max(1.0, np.abs(cn).max()), (name, b)
            T.append(t); Cv.append(c)
        out[f"compose_{name}_t"], out[f"compose_{name}_cov"] = np.array(T), np.array(Cv)
    for tf in (0, 1):
        P, D = [], []
        for b in range(B):
            post = np.r_[s["prev"][b], np.zeros(24)]
            po, de = o.dead_reckon_pose(s["u"][b], s["velcov"], s["prev"][b], post, tf)
            pn, dn = npc.dead_reckon_pose(s["u"][b], s["velcov"], s["prev"][b], post, tf)
            assert same_transform(po[:7], pn[:7]) <= TOL and np.abs(po[7:] - pn[7:]).max() <= TOL and np.abs(de - dn).max() <= TOL
            P.append(po); D.append(de)
        out[f"dr_pose_tf{tf}_post"], out[f"dr_pose_tf{tf}_delta"] = np.array(P), np.array(D)
    objs = [o.AdaptiveAttitudeCov(s["m1"], s["m2"], s["gamma"], s["r2count"]) for _ in range(B)]
    twin = [npc.AdaptiveAttitudeCov(s["m1"], s["m2"], s["gamma"], s["r2count"]) for _ in range(B)]
    Rs = np.zeros((s["steps"], B, 3, 3))
    for k in range(s["steps"]):
        for b in range(B):
            Rs[k, b] = objs[b].matrix(s["xk"][k, b], s["Pk"][b], s["z"][k, b], s["H"][k, b], s["R"])
            Rn = twin[b].matrix(s["xk"][k, b], s["Pk"][b], s["z"][k, b], s["H"][k, b], s["R"])
            assert rel(Rs[k, b], Rn) <= 1e-11 and objs[b].r2.value == twin[b].r2count, (k, b)
    out["adaptive_R"] = Rs
    np.savez(os.path.join(OUT, "pose_ops.npz"), **out)


if __name__ == "__main__":
    msckf_ekf()
    dead_reckon()
    usckf_unit_test()
    usckf_spd()
    msckf_unit_test()
    msckf_batch()
    pose_ops()
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)))
