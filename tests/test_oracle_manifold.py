"""G1: the manifold identities the reference itself asserts (test/MsckfUnitTest.cpp:61-113),
restated against the CPU oracle, plus SO(3) exp/log known answers."""
import numpy as np
import pytest

from oracle import oracle as o
import scenarios as sc


def eq_state(lay, a, b):
    """MtkDynamicWrap::operator== (MtkWrap.hpp:231-235): (a - b).isZero(1e-12)."""
    return bool(np.all(np.abs(o.boxminus(lay, a, b)) <= 1e-12))


@pytest.mark.parametrize("k", [0, 4])
def test_states_case(k):
    # MsckfUnitTest.cpp:50-74 (STATES); the reference uses a default-constructed state (k = 0 clones)
    lay = o.layout(o.MULTI, k)
    x = o.identity_state(lay)
    v = o.vectorize(lay, x)
    assert o.dof(lay) == v.size == 12 + 6 * k                       # :61
    assert eq_state(lay, x, x)                                      # :62
    xb = o.set_from_vector(lay, v)                                  # :65
    assert eq_state(lay, x, xb)                                     # :66
    assert o.dof(o.layout(o.SINGLE)) == 12
    # ReducedState DOF == 6 (:71): pos + orient -- same tangent size as SensorState
    assert o.dof(o.layout(o.MULTI, 1)) - o.dof(o.layout(o.MULTI, 0)) == 6


@pytest.mark.parametrize("k", [0, 3])
def test_operations_case(k):
    # MsckfUnitTest.cpp:76-120 (OPERATIONS)
    lay = o.layout(o.MULTI, k)
    mstate = o.identity_state(lay)
    mstatebis = o.identity_state(lay)
    mstatebis[0:3] = [1.0, 2.0, -3.0]                                # :79
    euler = np.full(3, 1.0 * sc.D2R)                                 # :82-85
    mstatebis[3:7] = o.quat_mul(mstatebis[3:7], o.so3_exp(euler))    # orient.boxplus(euler) :87
    for c in range(k):                                               # :89-94
        mstatebis[13 + 7 * c:13 + 7 * c + 3] = mstatebis[0:3]
        mstatebis[13 + 7 * c + 3:13 + 7 * c + 7] = mstatebis[3:7]
    vres = o.boxminus(lay, mstate, mstatebis)                        # :104
    resstate = o.set_from_vector(lay, vres)                          # :106
    sumstate = o.boxplus(lay, mstate, vres)                          # :109
    assert eq_state(lay, resstate, sumstate)                         # :110
    sumstate = o.boxplus(lay, mstate, -vres)                         # :111-112
    assert eq_state(lay, mstatebis, sumstate)                        # :113


def test_exp_log_known_answers():
    # rotation by pi/2 about z: q = (0, 0, sin(pi/4), cos(pi/4))
    q = o.so3_exp([0, 0, np.pi / 2])
    np.testing.assert_allclose(q, [0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)], atol=1e-16)
    np.testing.assert_allclose(o.so3_log(q), [0, 0, np.pi / 2], atol=1e-15)
    # identity and the tiny-angle Taylor branch of MTK cos_sinc_sqrt
    np.testing.assert_allclose(o.so3_exp([0, 0, 0]), [0, 0, 0, 1], atol=0)
    v = np.array([1e-3, -2e-3, 5e-4])
    th = np.linalg.norm(v)
    np.testing.assert_allclose(o.so3_exp(v), np.r_[np.sin(th / 2) / th * v, np.cos(th / 2)], rtol=1e-15, atol=1e-18)
    np.testing.assert_allclose(o.so3_log(o.so3_exp(v)), v, rtol=1e-13)
    # log uses atan (not atan2): q and -q map to the same tangent vector
    q = o.so3_exp([0.3, -0.2, 0.9])
    np.testing.assert_allclose(o.so3_log(-q), o.so3_log(q), rtol=1e-15)


def test_quat_rotate_matches_matrix():
    rng = np.random.default_rng(1)
    for _ in range(10):
        q = o.so3_exp(rng.normal(size=3))
        v = rng.normal(size=3)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(o.quat_rotate(q, v), R @ v, rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("kind,kw", [(o.SINGLE, {}), (o.MULTI, dict(k=5)), (o.AUGMENTED, dict(nfk=3, nfkl=9))])
def test_boxplus_boxminus_roundtrip(kind, kw):
    lay = o.layout(kind, **kw)
    rng = np.random.default_rng(2)
    x = o.set_from_vector(lay, rng.normal(size=o.dof(lay)))
    v = 0.4 * rng.normal(size=o.dof(lay))
    y = o.boxplus(lay, x, v)
    np.testing.assert_allclose(o.boxminus(lay, y, x), v, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(o.boxminus(lay, o.boxplus(lay, x, np.zeros_like(v)), x), 0, atol=1e-15)
