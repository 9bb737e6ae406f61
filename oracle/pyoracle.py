"""ctypes front-end of the CPU oracle (oracle/_build/libslam_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never from the conan_slam_amd package.  PARITY UNPINNED (see slam_oracle.h).

All matrices are numpy arrays in Fortran (column-major) order, as the reference's Eigen objects are.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLAM_ORACLE_LIB: an alternative build of the same sources (e.g. `make -C oracle asan` for the sanitizer run)
_LIB_PATH = os.environ.get("SLAM_ORACLE_LIB") or os.path.join(_HERE, "_build", "libslam_oracle.so")

Q_LOWER_CHOL_GAIN = 1
Q_PREDICT_NM4 = 2
REF_EXACT = Q_LOWER_CHOL_GAIN | Q_PREDICT_NM4
TEXTBOOK = 0

CHOL_OK, CHOL_EIGEN, CHOL_ZEROED = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only, no external dependencies)."""
    srcs = ["slam_oracle.c", "slam_oracle_impl.inc", "slam_oracle_pf.inc", "slam_oracle_fast.c", "slam_oracle.h"]
    stale = force or not os.path.exists(_LIB_PATH)
    if not stale:
        t = os.path.getmtime(_LIB_PATH)
        stale = any(os.path.getmtime(os.path.join(_HERE, s)) > t for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One precision flavour of the oracle: Oracle(np.float32) or Oracle(np.float64)."""

    def __init__(self, dtype=np.float32, quirks: int = REF_EXACT):
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.float32), np.dtype(np.float64))
        self.suf = "f32" if self.dtype == np.float32 else "f64"
        self.ct = C.c_float if self.dtype == np.float32 else C.c_double
        self.quirks = quirks
        self.L = lib()

    # ---- helpers
    def _f(self, name):
        return getattr(self.L, f"{name}_{self.suf}")

    def arr(self, a, order="F"):
        return np.array(a, dtype=self.dtype, order=order)

    def _chk(self, a: np.ndarray, ndim=None):
        assert a.dtype == self.dtype, (a.dtype, self.dtype)
        if a.ndim == 2:
            assert a.flags.f_contiguous or a.shape[0] == 1 or a.shape[1] == 1
        else:
            assert a.flags.c_contiguous

    # ---- scalar / small helpers
    def pi2pi(self, angle):
        f = self._f("orc_pi2pi")
        f.restype = self.ct
        f.argtypes = [self.ct]
        return self.dtype.type(f(self.ct(float(angle))))

    def cholesky_decomposition(self, M):
        M = self.arr(M)
        k = M.shape[0]
        L = np.zeros((k, k), dtype=self.dtype, order="F")
        f = self._f("orc_cholesky_decomposition")
        f.restype = C.c_int
        code = f(_ptr(M), C.c_int(k), _ptr(L))
        return L, code

    def inverse(self, A):
        A = self.arr(A)
        k = A.shape[0]
        out = np.zeros((k, k), dtype=self.dtype, order="F")
        self._f("orc_inverse")(_ptr(A), C.c_int(k), _ptr(out))
        return out

    def gain_factor(self, S, quirks=None):
        S = self.arr(S)
        k = S.shape[0]
        G = np.zeros((k, k), dtype=self.dtype, order="F")
        f = self._f("orc_gain_factor")
        f.restype = C.c_int
        code = f(_ptr(S), C.c_int(k), C.c_int(self.quirks if quirks is None else quirks), _ptr(G))
        return G, code

    # ---- EKF path: X (n,), P (cap,cap) Fortran; operate in place on the leading n x n
    def cholesky_update(self, X, P, n, V, R, H, quirks=None):
        self._chk(X), self._chk(P)
        V, R, H = self.arr(V), self.arr(R), self.arr(H)
        k = V.shape[0]
        f = self._f("orc_cholesky_update")
        f.restype = C.c_int
        return f(_ptr(X), _ptr(P), C.c_int(n), C.c_int(P.shape[0]), _ptr(V), _ptr(R), _ptr(H), C.c_int(k),
                 C.c_int(self.quirks if quirks is None else quirks))

    def joseph_update(self, X, P, n, V, R, H):
        self._chk(X), self._chk(P)
        V, R, H = self.arr(V), self.arr(R), self.arr(H)
        k = V.shape[0]
        self._f("orc_joseph_update")(_ptr(X), _ptr(P), C.c_int(n), C.c_int(P.shape[0]), _ptr(V), _ptr(R), _ptr(H),
                                     C.c_int(k))

    def observe_model(self, X, n, idf):
        self._chk(X)
        Zp = np.zeros(2, dtype=self.dtype)
        H = np.zeros((2, n), dtype=self.dtype, order="F")
        self._f("orc_ekf_observe_model")(_ptr(X), C.c_int(n), C.c_int(int(idf)), _ptr(Zp), _ptr(H))
        return Zp, H

    def predict(self, X, P, n, v, swa, Q, wb, dt, quirks=None):
        self._chk(X), self._chk(P)
        Q = self.arr(Q)
        f = self._f("orc_ekf_predict")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, self.ct, self.ct, C.c_void_p, self.ct, self.ct, C.c_int]
        f(_ptr(X), _ptr(P), n, P.shape[0], float(v), float(swa), _ptr(Q), float(wb), float(dt),
          self.quirks if quirks is None else quirks)

    def update(self, X, P, n, Z, R, idf, batch=True, quirks=None, fast=False):
        self._chk(X), self._chk(P)
        Z = self.arr(Z).reshape(2, -1, order="F") if np.size(Z) else np.zeros((2, 0), self.dtype, order="F")
        R = self.arr(R)
        idf = np.ascontiguousarray(idf, dtype=np.int32)
        m = Z.shape[1]
        q = self.quirks if quirks is None else quirks
        if fast:
            assert batch
            f = self._f("orc_ekf_batch_update_fast")
            f.restype = C.c_int
            return f(_ptr(X), _ptr(P), C.c_int(n), C.c_int(P.shape[0]), _ptr(Z), C.c_int(m), _ptr(R), _ptr(idf),
                     C.c_int(q))
        f = self._f("orc_ekf_update")
        f.restype = C.c_int
        return f(_ptr(X), _ptr(P), C.c_int(n), C.c_int(P.shape[0]), _ptr(Z), C.c_int(m), _ptr(R), _ptr(idf),
                 C.c_int(1 if batch else 0), C.c_int(q))

    def augment(self, X, P, n, Z, R):
        self._chk(X), self._chk(P)
        Z = self.arr(Z).reshape(2, -1, order="F") if np.size(Z) else np.zeros((2, 0), self.dtype, order="F")
        R = self.arr(R)
        q = Z.shape[1]
        assert X.shape[0] >= n + 2 * q and P.shape[0] >= n + 2 * q
        f = self._f("orc_ekf_augment")
        f.restype = C.c_int
        return f(_ptr(X), _ptr(P), C.c_int(n), C.c_int(P.shape[0]), _ptr(Z), C.c_int(q), _ptr(R))

    def observe_heading(self, X, P, n, phi, use=True, structured=False):
        self._chk(X), self._chk(P)
        f = self._f("orc_ekf_observe_heading_structured" if structured else "orc_ekf_observe_heading")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, self.ct, C.c_int]
        f(_ptr(X), _ptr(P), n, P.shape[0], float(phi), 1 if use else 0)

    def compute_association(self, X, P, n, z, R, idf):
        """EKF.cpp:131-144 -> (nis, nd)"""
        self._chk(X), self._chk(P)
        z, R = self.arr(z), self.arr(R, order="F")
        out = np.zeros(2, dtype=self.dtype)
        f = self._f("orc_ekf_compute_association")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        f(_ptr(X), _ptr(P), n, P.shape[0], _ptr(z), _ptr(R), int(idf), _ptr(out))
        return out[0], out[1]

    def data_associate(self, X, P, n, Z, R, gate1, gate2):
        """EKF.cpp:235-326 -> (idf[m] 1-based or 0, kind[m]: 0 dropped / 1 associated / 2 new feature)"""
        self._chk(X), self._chk(P)
        Z = self.arr(Z, order="F")
        R = self.arr(R, order="F")
        m = Z.shape[1] if Z.ndim == 2 else Z.size // 2
        idf = np.zeros(m, dtype=np.int32)
        kind = np.zeros(m, dtype=np.int32)
        f = self._f("orc_ekf_data_associate")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, self.ct, self.ct,
                      C.c_void_p, C.c_void_p]
        f(_ptr(X), _ptr(P), n, P.shape[0], _ptr(Z), m, _ptr(R), float(gate1), float(gate2), _ptr(idf), _ptr(kind))
        return idf, kind

    # ---- simulator helpers
    def vehicle_model(self, Xv, v, swa, wb, dt):
        self._chk(Xv)
        f = self._f("orc_vehicle_model")
        f.argtypes = [C.c_void_p, self.ct, self.ct, self.ct, self.ct]
        f(_ptr(Xv), float(v), float(swa), float(wb), float(dt))

    def compute_swa(self, Xv, WP, iwp, minD, swa, rateSWA, maxSWA, dt, int_signum=True):
        self._chk(Xv)
        WP = self.arr(WP)
        ciwp = C.c_int(int(iwp))
        cswa = self.ct(float(swa))
        f = self._f("orc_compute_swa")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), self.ct, C.POINTER(self.ct), self.ct,
                      self.ct, self.ct, C.c_int]
        f(_ptr(Xv), _ptr(WP), WP.shape[1], C.byref(ciwp), float(minD), C.byref(cswa), float(rateSWA), float(maxSWA),
          float(dt), 1 if int_signum else 0)
        return ciwp.value, self.dtype.type(cswa.value)

    def get_observations(self, Xv, LM, rmax):
        self._chk(Xv)
        LM = self.arr(LM)
        nlm = LM.shape[1]
        Z = np.zeros((2, nlm), dtype=self.dtype, order="F")
        tags = np.zeros(nlm, dtype=np.int32)
        f = self._f("orc_get_observations")
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, self.ct, C.c_void_p, C.c_void_p]
        cnt = f(_ptr(Xv), _ptr(LM), nlm, float(rmax), _ptr(Z), _ptr(tags))
        return np.asfortranarray(Z[:, :cnt]), tags[:cnt].copy()

    def data_associate_table(self, Z, tags, table, nf):
        Z = self.arr(Z).reshape(2, -1, order="F")
        m = Z.shape[1]
        tags = np.ascontiguousarray(tags, dtype=np.int32)
        assert table.dtype == np.int32
        ZF = np.zeros((2, max(m, 1)), dtype=self.dtype, order="F")
        ZN = np.zeros((2, max(m, 1)), dtype=self.dtype, order="F")
        idf = np.zeros(max(m, 1), dtype=np.int32)
        mn = C.c_int(0)
        f = self.L.orc_data_associate_table
        f.restype = C.c_int
        mf = f(_ptr(Z), _ptr(tags), C.c_int(m), _ptr(table), C.c_int(int(nf)), _ptr(ZF), _ptr(idf), _ptr(ZN),
               C.byref(mn), C.c_int(self.dtype.itemsize))
        return np.asfortranarray(ZF[:, :mf]), np.asfortranarray(ZN[:, :mn.value]), idf[:mf].copy()

    # ---- particle-filter path (per particle)
    def pf_predict(self, Xv, Pv, v, swa, Q, wb, dt):
        self._chk(Xv), self._chk(Pv)
        Q = self.arr(Q)
        f = self._f("orc_pf_predict")
        f.argtypes = [C.c_void_p, C.c_void_p, self.ct, self.ct, C.c_void_p, self.ct, self.ct]
        f(_ptr(Xv), _ptr(Pv), float(v), float(swa), _ptr(Q), float(wb), float(dt))

    def pf_observe_heading(self, Xv, Pv, phi, use=True):
        f = self._f("orc_pf_observe_heading")
        f.argtypes = [C.c_void_p, C.c_void_p, self.ct, C.c_int]
        f(_ptr(Xv), _ptr(Pv), float(phi), 1 if use else 0)

    def pf_compute_jacobians(self, Xv, XF, PF, idf, R):
        idf = np.ascontiguousarray(idf, dtype=np.int32)
        R = self.arr(R)
        ln = idf.shape[0]
        ZP = np.zeros((2, ln), self.dtype, order="F")
        HV = np.zeros((6, ln), self.dtype, order="F")
        HF = np.zeros((4, ln), self.dtype, order="F")
        SF = np.zeros((4, ln), self.dtype, order="F")
        self._f("orc_pf_compute_jacobians")(_ptr(Xv), _ptr(XF), _ptr(PF), _ptr(idf), C.c_int(ln), _ptr(R), _ptr(ZP),
                                            _ptr(HV), _ptr(HF), _ptr(SF))
        return ZP, HV, HF, SF

    def pf_gauss_evaluate(self, V, S, log_flag=False):
        V, S = self.arr(V), self.arr(S)
        f = self._f("orc_pf_gauss_evaluate")
        f.restype = self.ct
        return self.dtype.type(f(_ptr(V), _ptr(S), C.c_int(V.shape[0]), C.c_int(1 if log_flag else 0)))

    def pf_sample_proposal(self, w, Xv, Pv, XF, PF, Z, idf, R, normals):
        """w: 1-element array (in/out)."""
        Z = self.arr(Z).reshape(2, -1, order="F")
        idf = np.ascontiguousarray(idf, dtype=np.int32)
        R, normals = self.arr(R), self.arr(normals)
        self._f("orc_pf_sample_proposal")(_ptr(w), _ptr(Xv), _ptr(Pv), _ptr(XF), _ptr(PF), _ptr(Z), _ptr(idf),
                                          C.c_int(Z.shape[1]), _ptr(R), _ptr(normals))

    def pf_feature_update(self, Xv, XF, PF, Z, idf, R, quirks=None):
        Z = self.arr(Z).reshape(2, -1, order="F")
        idf = np.ascontiguousarray(idf, dtype=np.int32)
        R = self.arr(R)
        self._f("orc_pf_feature_update")(_ptr(Xv), _ptr(XF), _ptr(PF), _ptr(Z), _ptr(idf), C.c_int(Z.shape[1]),
                                         _ptr(R), C.c_int(self.quirks if quirks is None else quirks))

    def pf_add_features(self, Xv, XF, PF, nf, Z, R):
        Z = self.arr(Z).reshape(2, -1, order="F")
        R = self.arr(R)
        f = self._f("orc_pf_add_features")
        f.restype = C.c_int
        return f(_ptr(Xv), _ptr(XF), _ptr(PF), C.c_int(nf), _ptr(Z), C.c_int(Z.shape[1]), _ptr(R))

    def pf_stratified_random(self, n, noise, ref_exact=False):
        noise = self.arr(noise)
        out = np.zeros(n, self.dtype)
        self._f("orc_pf_stratified_random")(C.c_int(n), _ptr(noise), C.c_int(1 if ref_exact else 0), _ptr(out))
        return out

    def pf_stratified_resample(self, w, select, ref_exact=False):
        w = self.arr(w).copy()
        select = self.arr(select)
        n = w.shape[0]
        keep = np.zeros(n, np.int32)
        f = self._f("orc_pf_stratified_resample")
        f.restype = self.ct
        neff = f(_ptr(w), C.c_int(n), _ptr(select), _ptr(keep), C.c_int(1 if ref_exact else 0))
        return keep, self.dtype.type(neff), w

    def pf_normalize_resample(self, w, n_effective, flag, select):
        """w normalised in place. Returns (neff, resampled, keep)."""
        assert w.dtype == self.dtype
        select = self.arr(select)
        n = w.shape[0]
        keep = np.zeros(n, np.int32)
        res = C.c_int(0)
        f = self._f("orc_pf_normalize_resample")
        f.restype = self.ct
        neff = f(_ptr(w), C.c_int(n), C.c_int(int(n_effective)), C.c_int(1 if flag else 0), _ptr(select), _ptr(keep),
                 C.byref(res))
        return self.dtype.type(neff), bool(res.value), keep
