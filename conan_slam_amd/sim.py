"""Device-side observation generator and known-association table -- the Python mirror of
`cslam_sim_*` (include/cslam.h), named after the reference functions it stands in for:
    get_observations        Slam::getObservations         slam.h:575-683 (+ computeRangeBearing slam.h:339-368)
    add_observation_noise   the driver's sensor noise     slam.h:168-178 (the N(0,1) draws are an input)
    data_associate_table    EKF::dataAssociateTable       EKF.cpp:146-233
The map, the table, the scan and its split live in HBM; `device_ptrs()` feeds `EKF.update_device` without a host copy.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import F32, F64, check


class Simulator:
    def __init__(self, LM, dtype=np.float32, device: int = -1):
        self._L = _capi.lib()
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        LM = np.asarray(LM, dtype=self.dtype, order="F").reshape(2, -1, order="F")
        self.n_landmarks = LM.shape[1]
        self._h = C.c_void_p()
        check(self._L.cslam_sim_create(LM.ctypes.data_as(C.c_void_p), C.c_int(self.n_landmarks),
                                       C.c_int(F32 if self.dtype == np.float32 else F64), C.c_int(device),
                                       C.byref(self._h)))
        self._m = 0

    def close(self):
        if self._h:
            self._L.cslam_sim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_observations(self, xv_true, rmax):
        """-> (Z 2 x m, tags m): visible landmarks in ascending tag order (1-based tags)."""
        xv = np.ascontiguousarray(xv_true, dtype=self.dtype).reshape(3)
        Z = np.zeros((2, max(self.n_landmarks, 1)), dtype=self.dtype, order="F")
        tags = np.zeros(max(self.n_landmarks, 1), dtype=np.int32)
        m = C.c_int(0)
        check(self._L.cslam_sim_get_observations(self._h, xv.ctypes.data_as(C.c_void_p), C.c_double(float(rmax)),
                                                 Z.ctypes.data_as(C.c_void_p), tags.ctypes.data_as(C.POINTER(C.c_int)),
                                                 C.byref(m)))
        self._m = m.value
        return np.asfortranarray(Z[:, :m.value]), tags[:m.value].copy()

    def add_observation_noise(self, R, normals):
        """Z[r][i] += normals[2i+r] * sqrt(R[r][r]) on the device-resident scan."""
        R = np.asarray(R, dtype=self.dtype, order="F")
        nz = np.ascontiguousarray(normals, dtype=self.dtype).reshape(-1)
        if nz.size < 2 * self._m:
            raise ValueError("need two N(0,1) draws per observation")
        check(self._L.cslam_sim_add_observation_noise(self._h, R.ctypes.data_as(C.c_void_p), nz.ctypes.data_as(C.c_void_p)))

    def data_associate_table(self, n_features):
        """-> (ZF 2 x mf, ZN 2 x mn, idf mf) for the scan of the last get_observations(); updates the table."""
        cap = max(self._m, 1)
        ZF = np.zeros((2, cap), dtype=self.dtype, order="F")
        ZN = np.zeros((2, cap), dtype=self.dtype, order="F")
        idf = np.zeros(cap, dtype=np.int32)
        mf, mn = C.c_int(0), C.c_int(0)
        check(self._L.cslam_sim_associate_table(self._h, C.c_int(int(n_features)), ZF.ctypes.data_as(C.c_void_p),
                                                idf.ctypes.data_as(C.POINTER(C.c_int)), C.byref(mf),
                                                ZN.ctypes.data_as(C.c_void_p), C.byref(mn)))
        return np.asfortranarray(ZF[:, :mf.value]), np.asfortranarray(ZN[:, :mn.value]), idf[:mf.value].copy()

    dataAssociateTable = data_associate_table
    getObservations = get_observations

    def device_ptrs(self):
        """dict of raw device addresses: ZF, idf, ZN (last split), Z, tags (last scan)."""
        p = [C.c_void_p() for _ in range(5)]
        check(self._L.cslam_sim_device_ptrs(self._h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(p[3]), C.byref(p[4])))
        return dict(zip(("ZF", "idf", "ZN", "Z", "tags"), [int(x.value or 0) for x in p]))

    @property
    def table(self):
        t = np.zeros(max(self.n_landmarks, 1), dtype=np.int32)
        check(self._L.cslam_sim_get_table(self._h, t.ctypes.data_as(C.POINTER(C.c_int))))
        return t[:self.n_landmarks]

    @table.setter
    def table(self, value):
        t = np.ascontiguousarray(value, dtype=np.int32)
        if t.size != self.n_landmarks:
            raise ValueError("table size")
        check(self._L.cslam_sim_set_table(self._h, t.ctypes.data_as(C.POINTER(C.c_int))))
