"""ctypes binding of the CPU ORACLE (oracle/slk_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  See slk_oracle.h for
the "parity unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def _host_tag():
    """-march=native code must not travel between machines: key the build on the host CPU."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as fh:
            lines = [l for l in fh if l.startswith(("model name", "flags"))][:2]
    except OSError:
        lines = []
    return hashlib.sha1("".join(lines).encode()).hexdigest()[:10]


_SO = os.path.join(_BUILD, "libslk_oracle_%s.so" % _host_tag())

SINGLE, MULTI, AUGMENTED = 0, 1, 2
STATEK, STATEK_L, STATEK_I = 1, 2, 3
OK, LLT_FAIL, MEAN_NOT_CONVERGED, SINGULAR = 0, 1, 2, 4


def build(force=False, opt="-O3"):
    """Compile the oracle with gcc (ISO C mode: no FMA contraction, no -ffast-math; fp64 stays IEEE)."""
    src = os.path.join(_HERE, "slk_oracle.c")
    hdr = os.path.join(_HERE, "slk_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    os.makedirs(_BUILD, exist_ok=True)
    cmd = ["gcc", opt, "-march=native", "-std=c99", "-fPIC", "-shared", "-o", _SO, src, "-lm"]
    subprocess.check_call(cmd)
    return _SO


class Layout(C.Structure):
    _fields_ = [("kind", C.c_int), ("k", C.c_int), ("nfk", C.c_int), ("nfkl", C.c_int)]


class _Msckf(C.Structure):
    _fields_ = [("lay", Layout), ("mean", C.POINTER(C.c_double)), ("P", C.POINTER(C.c_double)),
                ("Fk", C.c_double * 144), ("mean_iters", C.c_int)]


class _Usckf(C.Structure):
    _fields_ = [("lay", Layout), ("mean", C.POINTER(C.c_double)), ("P", C.POINTER(C.c_double)),
                ("mean_iters", C.c_int)]


class ConstVelocity(C.Structure):
    _fields_ = [("velocity", C.c_double * 3), ("angular_velocity", C.c_double * 3), ("dt", C.c_double)]


class DeltaPose(C.Structure):
    _fields_ = [("dpos", C.c_double * 3), ("dquat", C.c_double * 4), ("velocity", C.c_double * 3),
                ("angular_velocity", C.c_double * 3)]


PROCESS_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
MEASURE_FN = C.CFUNCTYPE(None, C.POINTER(Layout), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L = _lib
        L.slko_dof.argtypes = [C.POINTER(Layout)]
        L.slko_storage.argtypes = [C.POINTER(Layout)]
        L.slko_so3_exp.argtypes = [dp, C.c_double, dp]
        L.slko_so3_log.argtypes = [dp, dp]
        L.slko_quat_mul.argtypes = [dp, dp, dp]
        L.slko_quat_rotate.argtypes = [dp, dp, dp]
        for n in ("slko_boxplus", "slko_boxminus"):
            getattr(L, n).argtypes = [C.POINTER(Layout), dp, dp, dp]
        L.slko_set_from_vector.argtypes = [C.POINTER(Layout), dp, dp]
        L.slko_vectorize.argtypes = [C.POINTER(Layout), dp, dp]
        L.slko_cholesky_lower.argtypes = [C.c_int, dp, dp]
        L.slko_inverse.argtypes = [C.c_int, dp, dp]
        L.slko_msckf_new.restype = C.POINTER(_Msckf)
        L.slko_msckf_new.argtypes = [C.c_int, dp, dp]
        L.slko_msckf_free.argtypes = [C.POINTER(_Msckf)]
        L.slko_msckf_predict.argtypes = [C.POINTER(_Msckf), C.c_void_p, C.c_void_p, dp]
        L.slko_msckf_update.argtypes = [C.POINTER(_Msckf), dp, C.c_int, C.c_void_p, C.c_void_p, dp, C.c_int,
                                        C.POINTER(C.c_uint)]
        L.slko_msckf_check_sigma_points.argtypes = [C.POINTER(_Msckf), dp, dp]
        L.slko_usckf_new_single.restype = C.POINTER(_Usckf)
        L.slko_usckf_new_single.argtypes = [dp, dp]
        L.slko_usckf_new.restype = C.POINTER(_Usckf)
        L.slko_usckf_new.argtypes = [C.c_int, C.c_int, dp, dp]
        L.slko_usckf_free.argtypes = [C.POINTER(_Usckf)]
        L.slko_usckf_cloning.argtypes = [C.POINTER(_Usckf), C.c_int]
        L.slko_usckf_set_measurement.argtypes = [C.POINTER(_Usckf), C.c_int, dp, C.c_int, dp]
        L.slko_usckf_predict.argtypes = [C.POINTER(_Usckf), C.c_void_p, C.c_void_p, dp]
        L.slko_usckf_update.argtypes = [C.POINTER(_Usckf), dp, C.c_int, C.c_void_p, C.c_void_p, dp, C.c_int,
                                        C.POINTER(C.c_int)]
        L.slko_accept_mahalanobis.argtypes = [C.c_double, C.c_int]
        L.slko_msckf_step_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, C.c_int,
                                            C.POINTER(C.c_uint)]
        L.slko_usckf_step_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _arr(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def _colmajor(M):
    """numpy [r, c] matrix -> flat column-major buffer."""
    return np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(-1)


def _from_colmajor(buf, r, c):
    return np.array(buf, dtype=np.float64).reshape(c, r).T.copy()


def fn_addr(name):
    return C.cast(getattr(lib(), name), C.c_void_p)


# ---------------------------------------------------------------- primitives
def layout(kind, k=0, nfk=0, nfkl=0):
    return Layout(kind, k, nfk, nfkl)


def dof(lay):
    return lib().slko_dof(C.byref(lay))


def storage(lay):
    return lib().slko_storage(C.byref(lay))


def so3_exp(v, scale=1.0):
    q = np.zeros(4)
    lib().slko_so3_exp(_p(_arr(v, 3)), scale, _p(q))
    return q


def so3_log(q):
    v = np.zeros(3)
    lib().slko_so3_log(_p(_arr(q, 4)), _p(v))
    return v


def quat_mul(a, b):
    o = np.zeros(4)
    lib().slko_quat_mul(_p(_arr(a, 4)), _p(_arr(b, 4)), _p(o))
    return o


def quat_rotate(q, v):
    o = np.zeros(3)
    lib().slko_quat_rotate(_p(_arr(q, 4)), _p(_arr(v, 3)), _p(o))
    return o


def boxplus(lay, x, v):
    out = np.zeros(storage(lay))
    lib().slko_boxplus(C.byref(lay), _p(_arr(x, storage(lay))), _p(_arr(v, dof(lay))), _p(out))
    return out


def boxminus(lay, a, b):
    out = np.zeros(dof(lay))
    lib().slko_boxminus(C.byref(lay), _p(_arr(a, storage(lay))), _p(_arr(b, storage(lay))), _p(out))
    return out


def set_from_vector(lay, v):
    out = np.zeros(storage(lay))
    lib().slko_set_from_vector(C.byref(lay), _p(_arr(v, dof(lay))), _p(out))
    return out


def vectorize(lay, x):
    out = np.zeros(dof(lay))
    lib().slko_vectorize(C.byref(lay), _p(_arr(x, storage(lay))), _p(out))
    return out


def cholesky_lower(A):
    n = A.shape[0]
    L = np.zeros(n * n)
    fail = lib().slko_cholesky_lower(n, _p(_colmajor(A)), _p(L))
    return _from_colmajor(L, n, n), fail


def inverse(A):
    n = A.shape[0]
    out = np.zeros(n * n)
    sing = lib().slko_inverse(n, _p(_colmajor(A)), _p(out))
    return _from_colmajor(out, n, n), sing


def identity_state(lay):
    x = np.zeros(storage(lay))
    if lay.kind == AUGMENTED:
        for s in range(3):
            x[13 * s + 6] = 1.0
    else:
        x[6] = 1.0
        if lay.kind == MULTI:
            for c in range(lay.k):
                x[13 + 7 * c + 6] = 1.0
    return x


# ---------------------------------------------------------------- models
class Model:
    """A (function pointer, context) pair handed to the oracle filters."""

    def __init__(self, fn, ctx_obj, ctx_ptr):
        self.fn, self._keep, self.ctx = fn, ctx_obj, ctx_ptr


def pm_const_velocity(velocity, angular_velocity, dt):
    s = ConstVelocity((C.c_double * 3)(*velocity), (C.c_double * 3)(*angular_velocity), dt)
    return Model(fn_addr("slko_pm_const_velocity"), s, C.cast(C.pointer(s), C.c_void_p))


def pm_delta_pose(dpos, dquat, velocity, angular_velocity):
    s = DeltaPose((C.c_double * 3)(*dpos), (C.c_double * 4)(*dquat), (C.c_double * 3)(*velocity),
                  (C.c_double * 3)(*angular_velocity))
    return Model(fn_addr("slko_pm_delta_pose"), s, C.cast(C.pointer(s), C.c_void_p))


def pm_dead_reckon(u):
    """DeadReckon::updatePose delta (src/core/DeadReckon.hpp:129-239) feeding the delta-pose model; u = dt v0 w0 v1 w1."""
    buf = (C.c_double * 13)(*[float(v) for v in u])
    return Model(fn_addr("slko_pm_dead_reckon"), buf, C.cast(buf, C.c_void_p))


def dead_reckon_delta(u):
    """[..., 13] inputs -> [..., 13] delta poses (dpos3 dquat4 velocity3 angular_velocity3)."""
    L = lib()
    L.slko_dead_reckon_delta.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.slko_dead_reckon_delta.restype = None
    uu = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape(-1, 13))
    out = np.zeros_like(uu)
    for r in range(uu.shape[0]):
        L.slko_dead_reckon_delta(uu[r].ctypes.data_as(C.POINTER(C.c_double)), out[r].ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(np.asarray(u).shape)


def pm_python(fn):
    """Opaque host functor (the reference's boost::bind form): fn(x13 ndarray) -> y13 ndarray."""
    def tramp(xp, yp, _ctx):
        x = np.ctypeslib.as_array(xp, shape=(13,)).copy()
        y = np.asarray(fn(x), dtype=np.float64)
        for i in range(13):
            yp[i] = y[i]
    cb = PROCESS_FN(tramp)
    return Model(C.cast(cb, C.c_void_p), cb, None)


def mm_vo_relative():
    return Model(fn_addr("slko_mm_vo_relative"), None, None)


def mm_feature_proj(features):
    a = _arr(features)
    return Model(fn_addr("slko_mm_feature_proj"), a, C.cast(_p(a), C.c_void_p))


def mm_pose_position(pose_index):
    a = np.array([float(pose_index)])
    return Model(fn_addr("slko_mm_pose_position"), a, C.cast(_p(a), C.c_void_p))


def mm_python(fn):
    """Opaque host functor: fn(X ndarray[Nq]) -> z ndarray[m]."""
    def tramp(layp, xp, m, zp, _ctx):
        nq = lib().slko_storage(layp)
        x = np.ctypeslib.as_array(xp, shape=(nq,)).copy()
        z = np.asarray(fn(x), dtype=np.float64)
        for i in range(m):
            zp[i] = z[i]
    cb = MEASURE_FN(tramp)
    return Model(C.cast(cb, C.c_void_p), cb, None)


# ---------------------------------------------------------------- filters
class Msckf:
    """localization::Msckf restated (reference src/filters/Msckf.hpp)."""

    def __init__(self, k, mean, P):
        self.lay = layout(MULTI, k)
        self.N, self.Nq = dof(self.lay), storage(self.lay)
        self._f = lib().slko_msckf_new(k, _p(_arr(mean, self.Nq)), _p(_colmajor(np.asarray(P).reshape(self.N, self.N))))

    def __del__(self):
        if getattr(self, "_f", None):
            lib().slko_msckf_free(self._f)
            self._f = None

    @property
    def mean(self):
        return np.ctypeslib.as_array(self._f.contents.mean, shape=(self.Nq,)).copy()

    @property
    def P(self):
        return _from_colmajor(np.ctypeslib.as_array(self._f.contents.P, shape=(self.N * self.N,)), self.N, self.N)

    @property
    def Fk(self):
        return _from_colmajor(np.array(self._f.contents.Fk), 12, 12)

    @property
    def mean_iters(self):
        return self._f.contents.mean_iters

    def predict(self, model, Q):
        return lib().slko_msckf_predict(self._f, model.fn, model.ctx, _p(_colmajor(Q)))

    def update(self, z, model, R, gate=True):
        z = _arr(z)
        no = C.c_uint(0)
        st = lib().slko_msckf_update(self._f, _p(z), z.size, model.fn, model.ctx, _p(_colmajor(R)), int(gate),
                                     C.byref(no))
        return st, no.value

    def update_ekf(self, z, zmean, H, R, gate=True):
        """EKF update (Msckf.hpp:284-349) with zmean = h(mu) and the m x N Jacobian H of the caller's functor."""
        z, zmean = _arr(z), _arr(zmean)
        m = z.size
        no = C.c_uint(0)
        L = lib()
        dp = C.POINTER(C.c_double)
        L.slko_msckf_update_ekf.argtypes = [C.POINTER(_Msckf), dp, dp, dp, C.c_int, dp, C.c_int, C.POINTER(C.c_uint)]
        Hc = _colmajor(np.asarray(H, dtype=np.float64).reshape(m, -1))
        st = L.slko_msckf_update_ekf(self._f, _p(z), _p(zmean), _p(Hc), m, _p(_colmajor(R)), int(gate), C.byref(no))
        return st, no.value

    def check_sigma_points(self):
        a, b = C.c_double(0), C.c_double(0)
        st = lib().slko_msckf_check_sigma_points(self._f, C.byref(a), C.byref(b))
        return st, a.value, b.value


class Usckf:
    """localization::Usckf restated (reference src/filters/Usckf.hpp)."""

    def __init__(self, state13=None, P0_12=None, nfk=0, nfkl=0, mean=None, P=None):
        if state13 is not None:
            self._f = lib().slko_usckf_new_single(_p(_arr(state13, 13)), _p(_colmajor(P0_12)))
        else:
            N = 36 + nfk + nfkl
            self._f = lib().slko_usckf_new(nfk, nfkl, _p(_arr(mean, 39 + nfk + nfkl)),
                                           _p(_colmajor(np.asarray(P).reshape(N, N))))

    def __del__(self):
        if getattr(self, "_f", None):
            lib().slko_usckf_free(self._f)
            self._f = None

    @property
    def lay(self):
        l = self._f.contents.lay
        return layout(l.kind, l.k, l.nfk, l.nfkl)

    @property
    def N(self):
        return dof(self.lay)

    @property
    def Nq(self):
        return storage(self.lay)

    @property
    def mean(self):
        return np.ctypeslib.as_array(self._f.contents.mean, shape=(self.Nq,)).copy()

    @property
    def P(self):
        N = self.N
        return _from_colmajor(np.ctypeslib.as_array(self._f.contents.P, shape=(N * N,)), N, N)

    @property
    def mean_iters(self):
        return self._f.contents.mean_iters

    def cloning(self, mode):
        lib().slko_usckf_cloning(self._f, mode)

    def set_measurement(self, mode, z, R):
        z = _arr(z)
        lib().slko_usckf_set_measurement(self._f, mode, _p(z), z.size, _p(_colmajor(R)))

    def predict(self, model, Q):
        return lib().slko_usckf_predict(self._f, model.fn, model.ctx, _p(_colmajor(Q)))

    def update(self, z, model, R, gate_dof=0):
        z = _arr(z)
        acc = C.c_int(0)
        st = lib().slko_usckf_update(self._f, _p(z), z.size, model.fn, model.ctx, _p(_colmajor(R)), gate_dof,
                                     C.byref(acc))
        return st, acc.value


def msckf_step_batch(k, m, steps, mean, P, u, feat, z, Q, R, gate=True):
    """In-place batch of `steps` x (predict + update); arrays are [B, ...] C-contiguous with each
    filter's P stored column-major (symmetric inputs make the order immaterial on entry)."""
    B = mean.shape[0]
    out = np.zeros(B, dtype=np.uint32)
    st = lib().slko_msckf_step_batch(B, k, m, steps, _p(mean), _p(P), _p(u), _p(feat), _p(z), _p(_colmajor(Q)),
                                     _p(_colmajor(R)), int(gate), out.ctypes.data_as(C.POINTER(C.c_uint)))
    return st, out


def usckf_step_batch(nfk, nfkl, steps, mean, P, u, z, Q, R):
    """In-place batch of `steps` x (Usckf predict + update), constant-velocity / relative-transform models."""
    return lib().slko_usckf_step_batch(mean.shape[0], nfk, nfkl, steps, _p(mean), _p(P), _p(u), _p(z), _p(_colmajor(Q)),
                                       _p(_colmajor(R)))


# ---------------------------------------------------------------- f3 / f4 batch helpers
def transform_compose(t2, cov2, t1, cov1):
    """TransformWithUncertainty::operator* (src/core/Transform.cpp:215-254): -> (t [7], cov [6, 6])."""
    L = lib()
    dp = C.POINTER(C.c_double)
    L.slko_transform_compose.argtypes = [dp, dp, dp, dp, dp, dp]
    L.slko_transform_compose.restype = None
    t2, t1 = _arr(t2, 7), _arr(t1, 7)
    c2 = _colmajor(cov2) if cov2 is not None else None
    c1 = _colmajor(cov1) if cov1 is not None else None
    out, oc = np.zeros(7), np.zeros(36)
    L.slko_transform_compose(_p(t2), _p(c2) if c2 is not None else None, _p(t1), _p(c1) if c1 is not None else None,
                             _p(out), _p(oc))
    return out, _from_colmajor(oc, 6, 6)


def update_pose_affine(prev, prev_cov, delta, delta_cov, use_tf):
    """DeadReckon::updatePose, Affine3d overload (src/core/DeadReckon.hpp:306-330)."""
    L = lib()
    dp = C.POINTER(C.c_double)
    L.slko_update_pose_affine.argtypes = [dp, dp, dp, dp, C.c_int, dp, dp]
    L.slko_update_pose_affine.restype = None
    out, oc = np.zeros(7), np.zeros(36)
    L.slko_update_pose_affine(_p(_arr(prev, 7)), _p(_colmajor(prev_cov)), _p(_arr(delta, 7)), _p(_colmajor(delta_cov)),
                              int(use_tf), _p(out), _p(oc))
    return out, _from_colmajor(oc, 6, 6)


def dead_reckon_pose(u, velcov, prev, post, use_tf):
    """DeadReckon::updatePose, RigidBodyState overload (src/core/DeadReckon.hpp:129-239): -> (post [49], delta [31])."""
    L = lib()
    dp = C.POINTER(C.c_double)
    L.slko_dead_reckon_pose.argtypes = [dp, dp, dp, dp, dp, C.c_int]
    L.slko_dead_reckon_pose.restype = None
    po, de = _arr(post, 49).copy(), np.zeros(31)
    L.slko_dead_reckon_pose(_p(_arr(u, 13)), _p(_colmajor(velcov)), _p(_arr(prev, 25)), _p(po), _p(de), int(use_tf))
    return po, de


class AdaptiveAttitudeCov:
    """AdaptiveAttitudeCov (src/filters/MeasurementModels.hpp:136-286), one object per filter like the reference."""

    def __init__(self, m1, m2, gamma, r2count):
        self.m1, self.m2, self.gamma = m1, m2, gamma
        self.hist = np.zeros(m1 * 9)
        self.r1 = C.c_uint(0)
        self.r2 = C.c_uint(r2count)

    def matrix(self, xk, Pk, z, H, R):
        L = lib()
        dp = C.POINTER(C.c_double)
        up = C.POINTER(C.c_uint)
        L.slko_adaptive_attitude_cov.argtypes = [C.c_uint, C.c_uint, C.c_double, dp, up, up, C.c_int, dp, dp, dp, dp, dp, dp]
        L.slko_adaptive_attitude_cov.restype = None
        n = len(xk)
        out = np.zeros(9)
        L.slko_adaptive_attitude_cov(self.m1, self.m2, self.gamma, _p(self.hist), C.byref(self.r1), C.byref(self.r2), n,
                                     _p(_arr(xk, n)), _p(_colmajor(np.asarray(Pk).reshape(n, n))), _p(_arr(z, 3)),
                                     _p(_colmajor(np.asarray(H).reshape(3, n))), _p(_colmajor(R)), _p(out))
        return _from_colmajor(out, 3, 3)
