"""Debug helper: one batched run per invocation.  usage: batch_probe.py quirks(t|r) n_inst N m steps calls(comma)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import test_batch_gpu as tb
from conan_slam_amd.synth import Workload
from pyoracle import REF_EXACT, TEXTBOOK

q = TEXTBOOK if sys.argv[1] == "t" else REF_EXACT
I, N, m, steps = (int(x) for x in sys.argv[2:6])
calls = tuple(int(x) for x in sys.argv[6].split(","))
loads = [Workload(N, m, np.float32, seed=300 + r, corr=0.02 if q == REF_EXACT else 0.5) for r in range(I)]
ctrl = [Workload(N, m, np.float32, seed=0, build_p=False).controls(t) for t in range(steps)]
inputs = tb._inputs(loads, steps)
print("start", flush=True)
states, tr, fl, nwin = tb._batch(loads, q, ctrl, inputs, steps, calls)
print("ok", fl, nwin, tr, flush=True)
