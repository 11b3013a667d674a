"""Host-side mirror of the reference filter interface on top of the C ABI (include/slk.h).

Class and method names follow the reference (src/filters/Msckf.hpp, Usckf.hpp): predict,
update, muState, muSingleState, getPk/setPk, PkAugmentedState, cloning, setMeasurement --
batched: every array carries a leading batch dimension B (B = 1 is the reference object).
Matrices are exchanged as numpy [B, rows, cols]; the column-major C-ABI layout is handled
here.  Python is plumbing only: every numerical operation runs in libslk_hip.so on the GPU
and there is NO CPU fallback -- loading or creating fails loudly without a HIP device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslk_hip.so")

MSCKF, USCKF = 1, 2
HOST, DEVICE = 0, 1
STATEK, STATEK_L, STATEK_I = 1, 2, 3
MODEL_EXTERNAL, PM_CONST_VELOCITY, PM_DELTA_POSE, PM_DEAD_RECKON = 0, 1, 2, 3
MM_VO_RELATIVE, MM_FEATURE_PROJ, MM_POSE_POSITION = 1, 2, 3
ST_LLT_FAIL, ST_MEAN_NOT_CONVERGED, ST_SINGULAR, ST_ALL_REJECTED, ST_EKF_ROWS, ST_BAD_INDEX = 1, 2, 4, 8, 16, 32
E_INVALID, E_NO_DEVICE, E_HIP, E_UNSUPPORTED, E_NOMEM = -1, -2, -3, -4, -5

# every symbol include/slk.h declares (checked by the CPU test-suite against the built library)
EXPORTS = [
    "slk_create", "slk_destroy", "slk_last_error", "slk_device_count", "slk_batch", "slk_dof", "slk_storage",
    "slk_set_state", "slk_get_state", "slk_mean_device_ptr", "slk_cov_device_ptr", "slk_predict", "slk_update",
    "slk_step", "slk_predict_sigma_points", "slk_predict_from_sigma", "slk_update_sigma_points",
    "slk_update_from_sigma", "slk_usckf_cloning", "slk_usckf_set_measurement", "slk_msckf_resize",
    "slk_get_outliers", "slk_get_status", "slk_clear_status", "slk_sync", "slk_timer_start", "slk_timer_stop",
    "slk_selftest_mfma", "slk_set_rebuild_precision", "slk_dead_reckon", "slk_msckf_clone_pose", "slk_msckf_drop_clone", "slk_update_ekf",
    "slk_check_sigma_points", "slk_update_innovation", "slk_update_selected", "slk_transform_compose",
    "slk_dead_reckon_pose", "slk_adaptive_create", "slk_adaptive_destroy", "slk_adaptive_matrix",
]


class SlkError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("kind", C.c_int), ("batch", C.c_int), ("device", C.c_int), ("n_clones", C.c_int),
                ("n_featuresk", C.c_int), ("n_featuresk_l", C.c_int), ("stream", C.c_void_p)]


_lib = None


def load_library(path=None):
    """dlopen libslk_hip.so and declare the prototypes.  Raises if the library is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SLK_HIP_LIB") or LIB_PATH      # SLK_HIP_LIB: A/B runs of experimental builds (tools/ab.sh)
    if not os.path.exists(p):
        raise SlkError(f"{p} not built: run `python slam-localization_amd/build.py` (needs hipcc)")
    lib = C.CDLL(p)
    vp, ip = C.c_void_p, C.c_int
    lib.slk_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.slk_destroy.argtypes = [vp]
    lib.slk_destroy.restype = None
    lib.slk_last_error.restype = C.c_char_p
    for n in ("slk_batch", "slk_dof", "slk_storage", "slk_clear_status", "slk_sync", "slk_timer_start"):
        getattr(lib, n).argtypes = [vp]
    lib.slk_set_state.argtypes = [vp, vp, vp, ip]
    lib.slk_get_state.argtypes = [vp, vp, vp, ip]
    lib.slk_mean_device_ptr.argtypes = [vp]
    lib.slk_mean_device_ptr.restype = vp
    lib.slk_cov_device_ptr.argtypes = [vp]
    lib.slk_cov_device_ptr.restype = vp
    lib.slk_predict.argtypes = [vp, ip, vp, ip, vp, ip, ip]
    lib.slk_dead_reckon.argtypes = [vp, vp, ip, vp, ip]
    lib.slk_update_ekf.argtypes = [vp, vp, vp, vp, ip, vp, ip, ip, ip]
    lib.slk_update.argtypes = [vp, ip, vp, ip, vp, ip, vp, ip, ip, ip]
    lib.slk_step.argtypes = [vp, ip, vp, ip, vp, ip, ip, vp, ip, vp, ip, vp, ip, ip, ip]
    lib.slk_predict_sigma_points.argtypes = [vp, vp, ip]
    lib.slk_predict_from_sigma.argtypes = [vp, vp, vp, ip, ip]
    lib.slk_update_sigma_points.argtypes = [vp, vp, ip]
    lib.slk_update_from_sigma.argtypes = [vp, vp, vp, ip, vp, ip, ip, ip]
    lib.slk_usckf_cloning.argtypes = [vp, ip]
    lib.slk_usckf_set_measurement.argtypes = [vp, ip, vp, ip, vp, ip]
    lib.slk_msckf_resize.argtypes = [vp, ip]
    lib.slk_msckf_clone_pose.argtypes = [vp]
    lib.slk_msckf_drop_clone.argtypes = [vp, ip]
    lib.slk_get_outliers.argtypes = [vp, vp, ip]
    lib.slk_get_status.argtypes = [vp, vp, ip]
    lib.slk_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.slk_selftest_mfma.argtypes = [ip]
    lib.slk_set_rebuild_precision.argtypes = [vp, ip]
    lib.slk_check_sigma_points.argtypes = [vp, vp, vp, ip]
    lib.slk_update_innovation.argtypes = [vp, ip, vp, ip, vp, vp, ip, vp, ip, vp, ip]
    lib.slk_update_selected.argtypes = [vp, ip, vp, ip, vp, vp, ip, vp, ip, vp, ip]
    lib.slk_transform_compose.argtypes = [vp, vp, vp, vp, vp, vp, vp, ip, ip]
    lib.slk_dead_reckon_pose.argtypes = [vp, vp, ip, vp, ip, vp, vp, vp, ip, ip]
    lib.slk_adaptive_create.argtypes = [ip, ip, C.c_uint, C.c_uint, C.c_double, C.c_uint, vp, C.POINTER(vp)]
    lib.slk_adaptive_destroy.argtypes = [vp]
    lib.slk_adaptive_destroy.restype = None
    lib.slk_adaptive_matrix.argtypes = [vp, ip, vp, vp, vp, vp, vp, ip, vp, ip]
    if path is None:
        _lib = lib
    return lib


def device_count():
    return load_library().slk_device_count()


def _check(rc, what):
    if rc != 0:
        msg = load_library().slk_last_error()
        raise SlkError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def _is_dev(a):
    return hasattr(a, "data_ptr")


class _Arg:
    """One marshalled argument: pointer, per-filter stride (0 = shared), location, keep-alive."""

    def __init__(self, ptr, stride, where, keep):
        self.ptr, self.stride, self.where, self.keep = ptr, stride, where, keep


def _rows(a, B, width):
    """Per-filter rows [B, >=width] or one shared row [>=width].  numpy -> host, torch cuda tensor -> device."""
    if a is None:
        return _Arg(None, 0, None, None)
    if _is_dev(a):
        assert a.is_contiguous()
        stride = 0 if a.dim() == 1 else int(a.stride(0))
        assert (a.shape[-1] if a.dim() > 1 else a.numel()) >= width
        return _Arg(a.data_ptr(), stride, DEVICE if a.is_cuda else HOST, a)
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim == 1:
        assert a.size >= width, (a.size, width)
        return _Arg(a.ctypes.data, 0, HOST, a)
    a = np.ascontiguousarray(a.reshape(a.shape[0], -1))
    assert a.shape[0] == B and a.shape[1] >= width, (a.shape, B, width)
    return _Arg(a.ctypes.data, a.shape[1], HOST, a)


def _zrows(z, B, m):
    """Measurement rows: the C ABI has no stride for z, slk_update / slk_step always read [B][m] doubles.  A single
    row is therefore broadcast to every filter here (numpy), device tensors must already hold B * m values."""
    if _is_dev(z):
        assert z.is_contiguous() and z.numel() == B * m, (tuple(z.shape), B, m)
        return _Arg(z.data_ptr(), m, DEVICE if z.is_cuda else HOST, z)
    z = np.asarray(z, dtype=np.float64)
    assert z.shape[-1] == m and (z.ndim == 1 or z.shape[0] in (1, B)), (z.shape, B, m)
    a = np.ascontiguousarray(np.broadcast_to(z.reshape(-1, m), (B, m)))
    return _Arg(a.ctypes.data, m, HOST, a)


def _mat(M, B, n):
    """n x n matrix, shared [n, n] or per-filter [B, n, n] (numpy: row/col indexable; torch device tensors
    must already be column-major per filter -- symmetric matrices are either way)."""
    if _is_dev(M):
        assert M.is_contiguous()
        stride = 0 if M.dim() == 2 else int(M.stride(0))
        return _Arg(M.data_ptr(), stride, DEVICE if M.is_cuda else HOST, M)
    M = np.asarray(M, dtype=np.float64)
    if M.ndim == 2:
        assert M.shape == (n, n), (M.shape, n)
        a = np.ascontiguousarray(M.T)
        return _Arg(a.ctypes.data, 0, HOST, a)
    assert M.shape == (B, n, n), (M.shape, B, n)
    a = np.ascontiguousarray(np.transpose(M, (0, 2, 1)))
    return _Arg(a.ctypes.data, n * n, HOST, a)


def _where(*args):
    ws = {a.where for a in args if a.where is not None}
    assert len(ws) == 1, "all arguments of one call must live on the same side (host or device)"
    return ws.pop()


def _np(model, m):
    return (m // 2) * 4 if model == MM_FEATURE_PROJ else (1 if model == MM_POSE_POSITION else 0)


class _FilterBatch:
    KIND = None

    def __init__(self, batch, device=0, stream=None, n_clones=0, nfk=0, nfkl=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        if self._lib.slk_device_count() <= 0:
            raise SlkError("no HIP device visible: slam-localization_amd has no CPU fallback")
        cfg = Config(self.KIND, batch, device, n_clones, nfk, nfkl, stream)
        _check(self._lib.slk_create(C.byref(cfg), C.byref(self._h)), "slk_create")
        self.B = batch

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.slk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- sizes
    @property
    def N(self):
        return self._lib.slk_dof(self._h)

    @property
    def Nq(self):
        return self._lib.slk_storage(self._h)

    def getDOF(self):                                         # State.hpp:373-376 / :590-593
        return self.N

    # ---- state
    def set_state(self, mean=None, P=None):
        N, Nq = self.N, self.Nq
        m = p = None
        if mean is not None:
            m = np.ascontiguousarray(np.broadcast_to(np.asarray(mean, dtype=np.float64), (self.B, Nq)))
        if P is not None:
            P = np.asarray(P, dtype=np.float64)
            P = np.broadcast_to(P, (self.B, N, N))
            p = np.ascontiguousarray(np.transpose(P, (0, 2, 1)))
        _check(self._lib.slk_set_state(self._h, m.ctypes.data if m is not None else None,
                                       p.ctypes.data if p is not None else None, HOST), "slk_set_state")

    def muState(self):                                        # Msckf.hpp:376-379 / Usckf.hpp:518-521
        m = np.empty((self.B, self.Nq))
        _check(self._lib.slk_get_state(self._h, m.ctypes.data, None, HOST), "slk_get_state")
        return m

    def _getP(self):
        N = self.N
        p = np.empty((self.B, N, N))
        _check(self._lib.slk_get_state(self._h, None, p.ctypes.data, HOST), "slk_get_state")
        return np.ascontiguousarray(np.transpose(p, (0, 2, 1)))

    def device_pointers(self):
        return self._lib.slk_mean_device_ptr(self._h), self._lib.slk_cov_device_ptr(self._h)

    def status(self):
        s = np.zeros(self.B, dtype=np.int32)
        _check(self._lib.slk_get_status(self._h, s.ctypes.data, HOST), "slk_get_status")
        return s

    def clear_status(self):
        _check(self._lib.slk_clear_status(self._h), "slk_clear_status")

    def outliers(self):
        o = np.zeros(self.B, dtype=np.uint32)
        _check(self._lib.slk_get_outliers(self._h, o.ctypes.data, HOST), "slk_get_outliers")
        return o

    def set_rebuild_precision(self, mode):
        """0 = fp64 (parity path), 1 = fp32 MFMA, 2 = bf16 operands / fp32 accumulation (precision sweep only)."""
        _check(self._lib.slk_set_rebuild_precision(self._h, int(mode)), "slk_set_rebuild_precision")

    def sync(self):
        _check(self._lib.slk_sync(self._h), "slk_sync")

    def timer_start(self):
        _check(self._lib.slk_timer_start(self._h), "slk_timer_start")

    def timer_stop(self):
        ms = C.c_float(0)
        _check(self._lib.slk_timer_stop(self._h, C.byref(ms)), "slk_timer_stop")
        return ms.value

    def _default_gate(self, gate):
        if gate is None:
            return 1 if self.KIND == MSCKF else 0             # Msckf.hpp:199 / Usckf.hpp:249
        return int(gate)

    # ---- Tier A: registered models
    def predict(self, model, u, Q):
        """predict(f, Q), f = registered process model `model` with inputs u (Msckf.hpp:89-95, Usckf.hpp:107-111)."""
        ua = _rows(u, self.B, 7 if model == PM_CONST_VELOCITY else 13)
        qa = _mat(Q, self.B, 12)
        _check(self._lib.slk_predict(self._h, model, ua.ptr, ua.stride, qa.ptr, qa.stride, _where(ua, qa)), "slk_predict")

    def update(self, z, model, params, R, gate=None):
        """update(z, h, R), h = registered measurement model (Msckf.hpp:196-213, Usckf.hpp:246-258)."""
        m = int(z.shape[-1])
        pa = _rows(params, self.B, _np(model, m)) if _np(model, m) else _Arg(None, 0, None, None)
        za, ra = _zrows(z, self.B, m), _mat(R, self.B, m)
        _check(self._lib.slk_update(self._h, model, pa.ptr, pa.stride, za.ptr, m, ra.ptr, ra.stride,
                                    self._default_gate(gate), _where(pa, za, ra)), "slk_update")

    def step(self, pmodel, u, Q, z, mmodel, params, R, gate=None):
        """predict + update fused into one launch (the benchmark's filter step)."""
        m = int(z.shape[-1])
        ua = _rows(u, self.B, 7 if pmodel == PM_CONST_VELOCITY else 13)
        qa = _mat(Q, self.B, 12)
        pa = _rows(params, self.B, _np(mmodel, m)) if _np(mmodel, m) else _Arg(None, 0, None, None)
        za, ra = _zrows(z, self.B, m), _mat(R, self.B, m)
        _check(self._lib.slk_step(self._h, pmodel, ua.ptr, ua.stride, qa.ptr, qa.stride, mmodel, pa.ptr, pa.stride,
                                  za.ptr, m, ra.ptr, ra.stride, self._default_gate(gate), _where(ua, qa, pa, za, ra)),
               "slk_step")

    # ---- Tier B: opaque host functors (the reference's boost::bind form)
    def predict_sigma_points(self):
        X = np.empty((self.B, 25, 13))
        _check(self._lib.slk_predict_sigma_points(self._h, X.ctypes.data, HOST), "slk_predict_sigma_points")
        return X

    def dead_reckon(self, u):
        """DeadReckon::updatePose delta poses (src/core/DeadReckon.hpp:129-239) for the batch: u = dt v0 w0 v1 w1
        (shared row or [B, 13]) -> [B, 13] = dpos dquat velocity angular_velocity, the `u` of PM_DELTA_POSE.
        numpy in -> numpy out; a torch device tensor in -> a torch device tensor out."""
        ua = _rows(u, self.B, 13)
        if ua.where == DEVICE:
            import torch
            out = torch.empty((self.B, 13), dtype=torch.float64, device=u.device)
            _check(self._lib.slk_dead_reckon(self._h, ua.ptr, ua.stride, out.data_ptr(), DEVICE), "slk_dead_reckon")
            return out
        out = np.empty((self.B, 13))
        _check(self._lib.slk_dead_reckon(self._h, ua.ptr, ua.stride, out.ctypes.data, HOST), "slk_dead_reckon")
        return out

    def transform_compose(self, t2, cov2, t1, cov1, additive=False):
        """TransformWithUncertainty::operator* (src/core/Transform.cpp:215-254) for the batch: t [B, 7] = pos quat,
        cov [B, 6, 6] in [r t] order or None (no uncertainty) -> (t [B, 7], cov [B, 6, 6]).  additive=True is the other
        branch of DeadReckon::updatePose's Affine3d overload (src/core/DeadReckon.hpp:317-323)."""
        B = self.B

        def cm(c):
            if c is None:
                return None
            return np.ascontiguousarray(np.transpose(np.broadcast_to(np.asarray(c, dtype=np.float64), (B, 6, 6)), (0, 2, 1)))
        a2 = np.ascontiguousarray(np.broadcast_to(np.asarray(t2, dtype=np.float64), (B, 7)))
        a1 = np.ascontiguousarray(np.broadcast_to(np.asarray(t1, dtype=np.float64), (B, 7)))
        c2, c1 = cm(cov2), cm(cov1)
        to, co = np.empty((B, 7)), np.empty((B, 6, 6))
        _check(self._lib.slk_transform_compose(self._h, a2.ctypes.data, c2.ctypes.data if c2 is not None else None,
                                               a1.ctypes.data, c1.ctypes.data if c1 is not None else None,
                                               to.ctypes.data, co.ctypes.data, int(bool(additive)), HOST), "slk_transform_compose")
        return to, np.ascontiguousarray(np.transpose(co, (0, 2, 1)))

    def dead_reckon_pose(self, u, velcov, prev, post, use_tf=False):
        """DeadReckon::updatePose, RigidBodyState overload (src/core/DeadReckon.hpp:129-239) for the batch.  Records as in
        include/slk.h: prev [B, 25], post [B, 49] (accumulated into without use_tf), -> (post [B, 49], delta [B, 31]);
        velcov [6, 6] shared or [B, 6, 6]."""
        B = self.B
        ua = _rows(u, B, 13)
        vc = np.asarray(velcov, dtype=np.float64)
        if vc.ndim == 2:
            vca, cs = np.ascontiguousarray(vc.T), 0
        else:
            vca, cs = np.ascontiguousarray(np.transpose(vc, (0, 2, 1))), 36
        pv = np.ascontiguousarray(np.broadcast_to(np.asarray(prev, dtype=np.float64), (B, 25)))
        po = np.array(np.broadcast_to(np.asarray(post, dtype=np.float64), (B, 49)), dtype=np.float64, order="C")
        de = np.empty((B, 31))
        _check(self._lib.slk_dead_reckon_pose(self._h, ua.ptr, ua.stride, vca.ctypes.data, cs, pv.ctypes.data, po.ctypes.data,
                                              de.ctypes.data, int(bool(use_tf)), HOST), "slk_dead_reckon_pose")
        return po, de

    def predict_functor(self, f, Q):
        """predict(f, Q) with an arbitrary Python callable f: state[13] -> state[13], applied on the host."""
        X = self.predict_sigma_points()
        Y = np.ascontiguousarray([[f(x) for x in Xb] for Xb in X], dtype=np.float64)
        qa = _mat(np.asarray(Q), self.B, 12)
        _check(self._lib.slk_predict_from_sigma(self._h, Y.ctypes.data, qa.ptr, qa.stride, HOST), "slk_predict_from_sigma")

    def update_sigma_points(self):
        X = np.empty((self.B, 2 * self.N + 1, self.Nq))
        _check(self._lib.slk_update_sigma_points(self._h, X.ctypes.data, HOST), "slk_update_sigma_points")
        return X

    def update_functor(self, z, h, R, gate=None):
        """update(z, h, R) with an arbitrary Python callable h: full state [Nq] -> z [m]."""
        X = self.update_sigma_points()
        Z = np.ascontiguousarray([[h(x) for x in Xb] for Xb in X], dtype=np.float64)
        m = Z.shape[-1]
        z = np.ascontiguousarray(np.broadcast_to(np.asarray(z, dtype=np.float64), (self.B, m)))
        ra = _mat(np.asarray(R), self.B, m)
        _check(self._lib.slk_update_from_sigma(self._h, Z.ctypes.data, z.ctypes.data, m, ra.ptr, ra.stride,
                                               self._default_gate(gate), HOST), "slk_update_from_sigma")


class Msckf(_FilterBatch):
    """Batched localization::Msckf<MultiState, State> (reference src/filters/Msckf.hpp)."""
    KIND = MSCKF

    def __init__(self, mean, P, device=0, stream=None):
        mean = np.atleast_2d(np.asarray(mean, dtype=np.float64))
        B, Nq = mean.shape
        k = (Nq - 13) // 7
        assert Nq == 13 + 7 * k, "mean must hold State(13) + k * SensorState(7)"
        super().__init__(B, device, stream, n_clones=k)
        self.set_state(mean, P)                              # Msckf(state, P0), Msckf.hpp:80-85

    def muSingleState(self):                                 # Msckf.hpp:356-361
        return self.muState()[:, :13]

    def getPk(self):                                         # Msckf.hpp:386-389
        return self._getP()

    def setPk(self, P):                                      # Msckf.hpp:391-395
        self.set_state(None, P)

    def getPkSingleState(self):                              # Msckf.hpp:368-374
        return self._getP()[:, :12, :12]

    def update_ekf(self, z, zmean, H, R, gate=True):
        """EKF update(z, h, H, R) (Msckf.hpp:284-349): zmean [B, m] = h(mu), H [B, m, N] = its Jacobian (numpy, row/col
        indexable), evaluated by the caller at the current mean like the reference's functor h(mu_state, H).
        Device tensors are taken as they are: z, zmean [B, m], H [B, N, m] (= m x N column-major per filter), R column-major."""
        if _is_dev(H):
            m = int(z.shape[-1])
            ra = _mat(R, self.B, m)
            assert z.is_contiguous() and zmean.is_contiguous() and H.is_contiguous()
            assert z.numel() == self.B * m and zmean.numel() == self.B * m and H.numel() == self.B * m * self.N
            _check(self._lib.slk_update_ekf(self._h, z.data_ptr(), zmean.data_ptr(), H.data_ptr(), m, ra.ptr, ra.stride,
                                            int(bool(gate)), DEVICE), "slk_update_ekf")
            return
        m = int(np.shape(z)[-1])
        z = np.ascontiguousarray(np.broadcast_to(np.asarray(z, dtype=np.float64).reshape(-1, m), (self.B, m)))
        zm = np.ascontiguousarray(np.broadcast_to(np.asarray(zmean, dtype=np.float64).reshape(-1, m), (self.B, m)))
        Hc = np.ascontiguousarray(np.transpose(np.asarray(H, dtype=np.float64).reshape(self.B, m, self.N), (0, 2, 1)))
        ra = _mat(np.asarray(R), self.B, m)
        _check(self._lib.slk_update_ekf(self._h, z.ctypes.data, zm.ctypes.data, Hc.ctypes.data, m, ra.ptr, ra.stride,
                                        int(bool(gate)), HOST), "slk_update_ekf")

    def checkSigmaPoints(self):
        """checkSigmaPoints() (Msckf.hpp:819-839) on the device: returns (max |covSigmaPoints - Pk| [B],
        |mu_state [-] muX| [B]); the reference asserts <= 1e-6 and == 0 (isZero(1e-12))."""
        ce, me = np.empty(self.B), np.empty(self.B)
        _check(self._lib.slk_check_sigma_points(self._h, ce.ctypes.data, me.ctypes.data, HOST), "slk_check_sigma_points")
        return ce, me

    def clone_pose(self):
        """Device-side muState().sensorsk.push_back(current pose) + setPk(J P J^T) (Msckf.hpp:381-395)."""
        _check(self._lib.slk_msckf_clone_pose(self._h), "slk_msckf_clone_pose")

    def drop_clone(self, index=0):
        """Device-side erase of clone `index` (0 = oldest) with its covariance rows / columns."""
        _check(self._lib.slk_msckf_drop_clone(self._h, int(index)), "slk_msckf_drop_clone")


class Usckf(_FilterBatch):
    """Batched localization::Usckf<AugmentedState, State> (reference src/filters/Usckf.hpp)."""
    KIND = USCKF

    def __init__(self, mean=None, P=None, nfk=0, nfkl=0, state_single=None, P0_single=None, device=0, stream=None):
        if state_single is not None:
            # Usckf(single_state, P0_single), Usckf.hpp:90-103: place the state, then clone twice
            s = np.atleast_2d(np.asarray(state_single, dtype=np.float64))
            B = s.shape[0]
            super().__init__(B, device, stream, nfk=0, nfkl=0)
            mean = np.zeros((B, 39))
            mean[:, [6, 19, 32]] = 1.0
            mean[:, 26:39] = s
            P = np.zeros((B, 36, 36))
            P[:, 24:36, 24:36] = P0_single
            self.set_state(mean, P)
            self.cloning(STATEK_I)
            self.cloning(STATEK_L)
        else:
            mean = np.atleast_2d(np.asarray(mean, dtype=np.float64))
            B = mean.shape[0]
            assert mean.shape[1] == 39 + nfk + nfkl
            super().__init__(B, device, stream, nfk=nfk, nfkl=nfkl)
            self.set_state(mean, P)                          # Usckf(state, P0), Usckf.hpp:83-86

    def cloning(self, mode):                                 # Usckf.hpp:391-433
        _check(self._lib.slk_usckf_cloning(self._h, int(mode)), "slk_usckf_cloning")

    def setMeasurement(self, mode, z, R):                    # Usckf.hpp:322-389
        z = np.atleast_1d(np.asarray(z, dtype=np.float64))
        n = z.shape[-1]
        zb = np.ascontiguousarray(np.broadcast_to(z, (self.B, n)))
        Rc = np.ascontiguousarray(np.asarray(R, dtype=np.float64).T)
        assert Rc.shape == (n, n)                            # assert (z_k_i.size() == R.rows()), :325-327
        _check(self._lib.slk_usckf_set_measurement(self._h, int(mode), zb.ctypes.data, n, Rc.ctypes.data, HOST),
               "slk_usckf_set_measurement")

    def muSingleState(self, which=STATEK_I):                 # Usckf.hpp:457-478
        o = {STATEK: 0, STATEK_L: 13, STATEK_I: 26}.get(which, 26)
        return self.muState()[:, o:o + 13]

    def PkAugmentedState(self):                              # Usckf.hpp:523-526
        return self._getP()

    def PkSingleState(self, which=STATEK_I):                 # Usckf.hpp:493-516
        o = {STATEK: 0, STATEK_L: 12, STATEK_I: 24}.get(which, 24)
        return self._getP()[:, o:o + 12, o:o + 12]


class AdaptiveAttitudeCov:
    """Batch of B independent localization::AdaptiveAttitudeCov objects (src/filters/MeasurementModels.hpp:136-286)
    resident on the device; matrix() is one call of the reference's ::matrix per object and returns the adapted
    measurement noise [B, 3, 3] -- the R of update()."""

    def __init__(self, batch, m1, m2, gamma, r2count, device=0, stream=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        _check(self._lib.slk_adaptive_create(batch, device, m1, m2, gamma, r2count, stream, C.byref(self._h)), "slk_adaptive_create")
        self.B = batch

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.slk_adaptive_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def matrix(self, xk, Pk, z, H, R):
        B = self.B
        x = np.ascontiguousarray(np.asarray(xk, dtype=np.float64).reshape(B, -1))
        n = x.shape[1]
        P = np.ascontiguousarray(np.transpose(np.asarray(Pk, dtype=np.float64).reshape(B, n, n), (0, 2, 1)))
        zz = np.ascontiguousarray(np.asarray(z, dtype=np.float64).reshape(B, 3))
        Hc = np.ascontiguousarray(np.transpose(np.asarray(H, dtype=np.float64).reshape(B, 3, n), (0, 2, 1)))
        ra = _mat(np.asarray(R), B, 3)
        out = np.empty((B, 3, 3))
        _check(self._lib.slk_adaptive_matrix(self._h, n, x.ctypes.data, P.ctypes.data, zz.ctypes.data, Hc.ctypes.data, ra.ptr,
                                             ra.stride, out.ctypes.data, HOST), "slk_adaptive_matrix")
        return np.ascontiguousarray(np.transpose(out, (0, 2, 1)))
