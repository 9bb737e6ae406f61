"""Step time as a function of the step index (deferred-128 headline loop): looks for drift over a long run."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conan_slam_amd import EKF, Q_TEXTBOOK
from conan_slam_amd.synth import Workload
sys.argv = [sys.argv[0]]
import bench

defer = int(os.environ.get("DEFER", "128"))
N, m, total = int(os.environ.get("LANDMARKS", "5000")), 32, 2100
dt = np.float64 if os.environ.get("DTYPE", "f32") == "f64" else np.float32
w = Workload(N, m, dt, seed=0)
eng = EKF(N, dtype=dt, quirks=Q_TEXTBOOK, sync_mode=False)
eng.set_state(w.X0, w.P0)
if defer:
    eng.set_deferred(defer)
inp = bench.DeviceInputs(torch, w, total)
def step(t):
    v, swa = inp.ctrl[t]
    eng.predict(v, swa, w.QE, w.wb, w.dt)
    eng.update_device(inp.z(t), m, w.RE, inp.i(t), batch=True)
for t in range(20):
    step(t)
eng.synchronize()
t = 20
for blk in range(20):
    t0 = time.perf_counter()
    h0 = time.perf_counter()
    for _ in range(100):
        step(t); t += 1
    host = time.perf_counter() - h0
    eng.synchronize()
    el = time.perf_counter() - t0
    print(f"steps {t-100:5d}-{t:5d}: {el/100*1e6:7.1f} us/step (host enqueue {host/100*1e6:6.1f} us/step) flags {eng.factor_status()} trace {eng.trace():.1f}")
