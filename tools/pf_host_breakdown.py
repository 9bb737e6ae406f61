"""Host-side time per call of one FastSLAM-2 observation step (512 x 1000, m = 8) -- where the step's time goes once the
kernels are small.  Usage (GPU box): python tools/pf_host_breakdown.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conan_slam_amd.pf import ParticleShard, SingleComm, resample_particles, stratified_random  # noqa: E402
from conan_slam_amd.synth import Workload, normal, uniform01  # noqa: E402

Np, Nf, m = 512, 1000, 8
dtype = np.float32
w = Workload(Nf, m, dtype, seed=0, build_p=False)
sh = ParticleShard(Np, Nf, dtype=dtype, n_global=Np)
XF = np.asfortranarray(np.stack([w.X0[3::2], w.X0[4::2]]).astype(dtype))
PF = np.asfortranarray(np.tile(np.array([1, 0, 0, 1], dtype=dtype)[:, None], (1, Nf)))
Pv = np.diag([0.05, 0.05, 1e-4]).astype(dtype)
for i in range(Np):
    sh.set_particle(i, 1.0 / Np, np.zeros(3, dtype), Pv, XF, PF)
comm = SingleComm()
acc = {}


def timed(name, fn):
    t0 = time.perf_counter()
    r = fn()
    sh.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r


steps = 200
for t in range(steps + 5):
    if t == 5:
        acc.clear()
    Z, idf = w.observations(t)
    nrm = np.ascontiguousarray(normal(500 + t, np.arange(3 * Np, dtype=np.uint64)).reshape(3, Np).astype(dtype))
    u = uniform01(900 + t, np.arange(Np, dtype=np.uint64))
    v, swa = w.controls(t)
    timed("predict", lambda: sh.predict(v, swa, w.QE, w.wb, w.dt))
    timed("sample_proposal", lambda: sh.sample_proposal(Z, idf, w.RE, nrm))
    timed("feature_update", lambda: sh.feature_update(Z, idf, w.RE))
    sel = timed("stratified_random (host numpy)", lambda: stratified_random(Np, u, dtype))
    timed("resample (device)", lambda: resample_particles(sh, comm, Np + 1, True, select=sel))
for k, v in acc.items():
    print(f"{k:34s} {v / steps * 1e6:8.1f} us/step (call + device time, synchronised)")
