"""Host-side mirror of the reference's EKF back-end over the C ABI (include/cslam.h).

`EKF` offers the reference's call surface for the hot path -- predict / update / augment /
observeHeading (slam.h:841-847, 938-943, 190-191, 788; EKF.cpp) -- with the same argument meaning:
feature indices are 1-based, Z is 2 x m (range; bearing), Q and R are 2 x 2.  The difference to the
reference is ownership: X and P live on the GPU inside the handle instead of being passed by reference
into every call; `X` / `P` properties (or get_state) download them.

All arithmetic happens in the HIP library; this file only marshals pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import (F32, F64, Q_REF_EXACT, Q_TEXTBOOK, check)


def _vp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class EKF:
    def __init__(self, max_landmarks: int, dtype=np.float32, device: int = -1, quirks: int = Q_REF_EXACT,
                 sync_mode: bool = True):
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        self._L = _capi.lib()
        self._h = C.c_void_p(None)
        code = F32 if self.dtype == np.float32 else F64
        check(self._L.cslam_ekf_create(C.c_int(max_landmarks), C.c_int(code), C.c_int(device), C.c_int(quirks),
                                       C.byref(self._h)))
        self.max_landmarks = max_landmarks
        self.quirks = quirks
        if not sync_mode:
            self.set_sync_mode(False)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.cslam_ekf_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ helpers
    def _arr(self, a, shape=None):
        a = np.ascontiguousarray(np.asarray(a, dtype=self.dtype).reshape(-1, order="F"))
        return a

    def _mat22(self, M):
        return np.asfortranarray(np.asarray(M, dtype=self.dtype).reshape(2, 2))

    def set_sync_mode(self, on: bool):
        check(self._L.cslam_ekf_set_sync_mode(self._h, C.c_int(1 if on else 0)))

    def set_deferred(self, max_pending_columns: int):
        """Keep up to this many W1 columns pending and apply them in one P-GEMM (0 = apply at once)."""
        check(self._L.cslam_ekf_set_deferred(self._h, C.c_int(int(max_pending_columns))))

    def flush(self):
        check(self._L.cslam_ekf_flush(self._h))

    def set_pgemm_workgroups(self, workgroups: int):
        """Cap the persistent P-GEMM grid (0 = whole chip): for instances that co-run on one GPU."""
        check(self._L.cslam_ekf_set_pgemm_workgroups(self._h, C.c_int(int(workgroups))))

    # ------------------------------------------------------------------ state
    @property
    def n(self) -> int:
        v = C.c_int(0)
        check(self._L.cslam_ekf_get_n(self._h, C.byref(v)))
        return v.value

    def set_state(self, X, P):
        X = np.ascontiguousarray(X, dtype=self.dtype)
        P = np.asfortranarray(P, dtype=self.dtype)
        n = X.shape[0]
        assert P.shape == (n, n)
        check(self._L.cslam_ekf_set_state(self._h, _vp(X), C.c_int(n), _vp(P), C.c_int(n)))

    def get_state(self):
        n = self.n
        X = np.zeros(n, dtype=self.dtype)
        P = np.zeros((n, n), dtype=self.dtype, order="F")
        check(self._L.cslam_ekf_get_state(self._h, _vp(X), _vp(P), C.c_int(n)))
        return X, P

    def get_x(self):
        n = self.n
        X = np.zeros(n, dtype=self.dtype)
        check(self._L.cslam_ekf_get_x(self._h, _vp(X), C.c_int(n)))
        return X

    def get_p(self):
        return self.get_state()[1]

    X = property(get_x)
    P = property(get_p)

    def trace(self) -> float:
        t = C.c_double(0.0)
        check(self._L.cslam_ekf_trace(self._h, C.byref(t)))
        return t.value

    def synchronize(self):
        check(self._L.cslam_ekf_synchronize(self._h))

    def factor_status(self, clear: bool = False) -> int:
        f = C.c_int(0)
        check(self._L.cslam_ekf_factor_status(self._h, C.byref(f), C.c_int(1 if clear else 0)))
        return f.value

    # ------------------------------------------------------------------ the hot path
    def predict(self, v, swa, Q, wb, dt):
        """Slam::predict(X, P, v, swa, Q, wb, dt) -- EKF.cpp:406-455."""
        Q = self._mat22(Q)
        check(self._L.cslam_ekf_predict(self._h, C.c_double(float(v)), C.c_double(float(swa)), _vp(Q),
                                        C.c_double(float(wb)), C.c_double(float(dt))))

    def update(self, Z, R, idf, batch: bool = False):
        """Slam::update(X, P, Z, R, idf, batch) -- EKF.cpp:481-496 (batch defaults to false, slam.h:943)."""
        Z = np.asarray(Z, dtype=self.dtype)
        m = 0 if Z.size == 0 else Z.reshape(2, -1, order="F").shape[1]
        Zc = self._arr(Z) if m else np.zeros(2, dtype=self.dtype)
        R = self._mat22(R)
        idf = np.ascontiguousarray(idf, dtype=np.int32)
        assert idf.shape[0] == m
        idp = idf.ctypes.data_as(C.c_void_p) if m else None
        check(self._L.cslam_ekf_update(self._h, _vp(Zc), C.c_int(m), _vp(R), idp, C.c_int(1 if batch else 0)))

    def update_device(self, dZ_ptr: int, m: int, R, d_idf_ptr: int, batch: bool = True):
        """update() with Z (2 x m scalars) and idf (m int32) already in HBM (raw device pointers)."""
        R = self._mat22(R)
        check(self._L.cslam_ekf_update_device(self._h, C.c_void_p(dZ_ptr), C.c_int(m), _vp(R), C.c_void_p(d_idf_ptr),
                                              C.c_int(1 if batch else 0)))

    def augment(self, Z, R):
        """Slam::augment(X, P, Z, R) -- EKF.cpp:9-26."""
        Z = np.asarray(Z, dtype=self.dtype)
        q = 0 if Z.size == 0 else Z.reshape(2, -1, order="F").shape[1]
        Zc = self._arr(Z) if q else np.zeros(2, dtype=self.dtype)
        R = self._mat22(R)
        check(self._L.cslam_ekf_augment(self._h, _vp(Zc), C.c_int(q), _vp(R)))

    def observe_heading(self, phi, use_heading: bool = False):
        """Slam::observeHeading(X, P, phi, useHeading) -- EKF.cpp:328-352 (default false, slam.h:788)."""
        check(self._L.cslam_ekf_observe_heading(self._h, C.c_double(float(phi)), C.c_int(1 if use_heading else 0)))

    def associate(self, Z, R, gate1, gate2):
        """Raw result of the gated nearest-neighbour search (EKF.cpp:235-326 with computeAssociation EKF.cpp:131-144):
        (idf[m], kind[m]); kind 1 = associated with the 1-based feature idf[i], 2 = new feature, 0 = dropped."""
        Z = np.asarray(Z, dtype=self.dtype, order="F").reshape(2, -1, order="F")
        R = np.asarray(R, dtype=self.dtype, order="F")
        m = Z.shape[1]
        idf = np.zeros(max(m, 1), dtype=np.int32)
        kind = np.zeros(max(m, 1), dtype=np.int32)
        check(self._L.cslam_ekf_associate(self._h, Z.ctypes.data_as(C.c_void_p), C.c_int(m), R.ctypes.data_as(C.c_void_p),
                                          C.c_double(float(gate1)), C.c_double(float(gate2)),
                                          idf.ctypes.data_as(C.POINTER(C.c_int)), kind.ctypes.data_as(C.POINTER(C.c_int))))
        return idf[:m], kind[:m]

    def data_associate(self, Z, R, gate1, gate2):
        """Slam::dataAssociate(X, P, Z, R, gate1, gate2) -> (ZF, ZN, idf) -- EKF.cpp:235-326.
        Under REF_EXACT quirks ZN is EMPTY, as in the reference (EKF.cpp:307 re-declares ZN inside the try block, so the
        returned matrix is the 0 x 0 one of line 243); TEXTBOOK returns the new-feature observations the loop found."""
        Z = np.asarray(Z, dtype=self.dtype, order="F").reshape(2, -1, order="F")
        idf, kind = self.associate(Z, R, gate1, gate2)
        ZF = np.asfortranarray(Z[:, kind == 1])
        if self.quirks == Q_REF_EXACT:
            ZN = np.zeros((0, 0), dtype=self.dtype)
        else:
            ZN = np.asfortranarray(Z[:, kind == 2])
        return ZF, ZN, idf[kind == 1].astype(np.int32)

    # reference-style aliases
    observeHeading = observe_heading
    dataAssociate = data_associate

    # ------------------------------------------------------------------ measurement / introspection
    def set_profiling(self, mode: int):
        """0 off, 1 every stage of update(), 2 every downdate (P-GEMM) launch, 3 one downdate launch in sixteen."""
        check(self._L.cslam_ekf_set_profiling(self._h, C.c_int(mode)))

    def stage_times(self):
        ms = (C.c_double * _capi.N_STAGES)()
        cnt = (C.c_int * _capi.N_STAGES)()
        check(self._L.cslam_ekf_get_stage_times(self._h, ms, cnt))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(_capi.STAGE_NAMES)}

    def debug_last_update(self):
        n = self.n
        k = C.c_int(0)
        check(self._L.cslam_ekf_debug_last_update(self._h, None, None, None, None, None, C.byref(k)))
        k = k.value
        out = {
            "PHT": np.zeros((n, k), self.dtype, order="F"),
            "S": np.zeros((k, k), self.dtype, order="F"),
            "G": np.zeros((k, k), self.dtype, order="F"),
            "W1": np.zeros((n, k), self.dtype, order="F"),
            "V": np.zeros(k, self.dtype),
        }
        if k:
            check(self._L.cslam_ekf_debug_last_update(self._h, _vp(out["PHT"]), _vp(out["S"]), _vp(out["G"]),
                                                      _vp(out["W1"]), _vp(out["V"]), C.byref(C.c_int(0))))
        return out


class EKFBatch:
    """`instances` independent f32 filters of `n_landmarks` landmarks each, advancing in lockstep (cslam_ekf_batch_*):
    the Monte-Carlo unit of BASELINE configs[4] (test/main.cpp:132-200 x I) with one launch per stage for all
    instances."""

    def __init__(self, instances: int, n_landmarks: int, device: int = -1, quirks: int = Q_REF_EXACT):
        self._L = _capi.lib()
        self._h = C.c_void_p(None)
        check(self._L.cslam_ekf_batch_create(C.c_int(instances), C.c_int(n_landmarks), C.c_int(device), C.c_int(quirks),
                                             C.byref(self._h)))
        self.instances, self.n_landmarks, self.n = instances, n_landmarks, 3 + 2 * n_landmarks
        self.quirks = quirks

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.cslam_ekf_batch_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_state(self, instance: int, X, P):
        X = np.ascontiguousarray(X, dtype=np.float32)
        P = np.asfortranarray(P, dtype=np.float32)
        if X.shape != (self.n,) or P.shape != (self.n, self.n):
            raise ValueError(f"state of {X.shape} / {P.shape}: the batch holds filters of n = {self.n}")
        check(self._L.cslam_ekf_batch_set_state(self._h, C.c_int(instance), _vp(X), C.c_int(self.n), _vp(P), C.c_int(self.n)))

    def get_state(self, instance: int):
        X = np.empty(self.n, dtype=np.float32)
        P = np.empty((self.n, self.n), dtype=np.float32, order="F")
        check(self._L.cslam_ekf_batch_get_state(self._h, C.c_int(instance), _vp(X), _vp(P), C.c_int(self.n)))
        return X, P

    def run(self, steps: int, v, swa, Q, wb: float, dt: float, dZ_ptrs, d_idf_ptrs, m: int, R):
        """steps x {predict; update} on every instance.  v / swa: `steps` controls (common to the instances);
        dZ_ptrs / d_idf_ptrs: one device pointer per instance (steps x 2m float32 / steps x m int32, step-major)."""
        v = np.ascontiguousarray(v, dtype=np.float64)
        swa = np.ascontiguousarray(swa, dtype=np.float64)
        if v.shape[0] < steps or swa.shape[0] < steps or len(dZ_ptrs) != self.instances or len(d_idf_ptrs) != self.instances:
            raise ValueError("run: controls shorter than `steps`, or not one input pointer per instance")
        Q = np.asfortranarray(Q, dtype=np.float32)
        R = np.asfortranarray(R, dtype=np.float32)
        zs = (C.c_void_p * self.instances)(*[int(p) for p in dZ_ptrs])
        ids = (C.c_void_p * self.instances)(*[int(p) for p in d_idf_ptrs])
        check(self._L.cslam_ekf_batch_run(self._h, C.c_int(steps), v.ctypes.data_as(C.POINTER(C.c_double)),
                                          swa.ctypes.data_as(C.POINTER(C.c_double)), _vp(Q), C.c_double(wb), C.c_double(dt),
                                          zs, ids, C.c_int(m), _vp(R)))

    def flush(self):
        check(self._L.cslam_ekf_batch_flush(self._h))

    def synchronize(self):
        check(self._L.cslam_ekf_batch_synchronize(self._h))

    def trace(self):
        tr = (C.c_double * self.instances)()
        check(self._L.cslam_ekf_batch_trace(self._h, tr))
        return [float(t) for t in tr]

    def factor_status(self):
        fl = (C.c_int * self.instances)()
        check(self._L.cslam_ekf_batch_factor_status(self._h, fl))
        return [int(f) for f in fl]

    def windows(self) -> int:
        w = C.c_longlong(0)
        check(self._L.cslam_ekf_batch_info(self._h, None, None, C.byref(w)))
        return w.value

    def set_profiling(self, every: int):
        check(self._L.cslam_ekf_batch_set_profiling(self._h, C.c_int(every)))

    def pgemm_time(self):
        """(sum of ms, launches) of the covariance-downdate launches timed since set_profiling(every > 0)."""
        ms, cnt = C.c_double(0.0), C.c_int(0)
        check(self._L.cslam_ekf_batch_get_pgemm_time(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value
