"""Timing of the device-side observation generator / association table at the BASELINE map size,
next to the oracle's host loops.  Usage (GPU box): python tools/sim_timing.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conan_slam_amd import Simulator  # noqa: E402
from pyoracle import Oracle  # noqa: E402

N = 5000
rng = np.random.default_rng(1)
LM = np.asfortranarray(rng.uniform(-400, 400, size=(2, N)).astype(np.float32))
sim = Simulator(LM)
o = Oracle(np.float32)
xv = np.array([10.0, -20.0, 0.4], dtype=np.float32)
table = np.zeros(N, dtype=np.int32)
Z, tags = sim.get_observations(xv, 100.0)
sim.data_associate_table(0)
iters = 500
t0 = time.perf_counter()
for i in range(iters):
    sim.get_observations(xv, 100.0)
    sim.data_associate_table(len(tags))
t_gpu = (time.perf_counter() - t0) / iters
t0 = time.perf_counter()
for i in range(iters):
    Zo, to = o.get_observations(xv, LM, 100.0)
    o.data_associate_table(Zo, to, table, len(to))
t_cpu = (time.perf_counter() - t0) / iters
print(json.dumps({"n_landmarks": N, "visible": int(len(tags)), "device_us_per_step_host_synchronous": t_gpu * 1e6,
                  "oracle_host_us_per_step": t_cpu * 1e6,
                  "note": "device path: 2 kernels + 3 small D2H copies with host sync per step (ctypes overhead included)"}))
