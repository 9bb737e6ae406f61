"""The batched Monte-Carlo engine (cslam_ekf_batch_*, BASELINE configs[4]: test/main.cpp:132-200 x I).

Every instance of a batch must end BITWISE equal to the same filter run alone through a single handle that executes the
same pairs of updates as look-ahead windows (CSLAM_LOOKAHEAD=1: the batched kernels call the same device functions), and
that filter is checked against the CPU oracle (slam.h:235-266 / EKF.cpp:93-129, 406-455) at SURVEY 8d's tolerances.

PARITY UNPINNED: the oracle is this build's CPU restatement; the reference holds no fixtures.
"""
import numpy as np
import pytest

from helpers import assert_close
from pyoracle import Oracle, REF_EXACT, TEXTBOOK

pytestmark = pytest.mark.gpu


def _inputs(loads, steps):
    import torch

    out = []
    for w in loads:
        m = w.m
        Zh = np.zeros((steps, 2 * m), dtype=np.float32)
        Ih = np.zeros((steps, m), dtype=np.int32)
        obs = []
        for t in range(steps):
            w.controls(t)  # (the vehicle of the workload moves with its own controls; the filters get the common ones)
            Z, idf = w.observations(t)
            Zh[t] = Z.reshape(-1, order="F")
            Ih[t] = idf
            obs.append((Z, idf))
        out.append((torch.from_numpy(Zh).cuda(), torch.from_numpy(Ih).cuda(), obs))
    torch.cuda.synchronize()
    return out


def _solo(w, quirks, ctrl, dZ, dI, steps):
    from conan_slam_amd import EKF

    eng = EKF(w.N, dtype=np.float32, quirks=quirks, sync_mode=False)
    eng.set_state(w.X0, w.P0)
    eng.set_deferred(128)
    for t in range(steps):
        v, swa = ctrl[t]
        eng.predict(v, swa, w.QE, w.wb, w.dt)
        eng.update_device(dZ.data_ptr() + t * 2 * w.m * 4, w.m, w.RE, dI.data_ptr() + t * w.m * 4, batch=True)
    X, P = eng.get_state()
    tr = eng.trace()
    fl = eng.factor_status()
    eng.close()
    return X, P, tr, fl


def _batch(loads, quirks, ctrl, inputs, steps, calls):
    from conan_slam_amd import EKFBatch

    w0 = loads[0]
    b = EKFBatch(len(loads), w0.N, quirks=quirks)
    for i, w in enumerate(loads):
        b.set_state(i, w.X0, w.P0)
    v = np.array([c[0] for c in ctrl], dtype=np.float64)
    s = np.array([c[1] for c in ctrl], dtype=np.float64)
    t0 = 0
    for cnt in calls:
        zs = [inp[0].data_ptr() + t0 * 2 * w0.m * 4 for inp in inputs]
        ids = [inp[1].data_ptr() + t0 * w0.m * 4 for inp in inputs]
        b.run(cnt, v[t0:], s[t0:], w0.QE, w0.wb, w0.dt, zs, ids, w0.m, w0.RE)
        t0 += cnt
    assert t0 == steps
    states = [b.get_state(i) for i in range(len(loads))]
    tr = b.trace()
    fl = b.factor_status()
    nwin = b.windows()
    b.close()
    return states, tr, fl, nwin


@pytest.mark.parametrize("quirks,n_inst,N,m,steps,calls", [
    (TEXTBOOK, 3, 300, 32, 6, (6,)),       # three windows of two updates
    (TEXTBOOK, 2, 300, 20, 5, (5,)),       # k = 40; the last window holds one update
    (REF_EXACT, 3, 300, 32, 4, (2, 2)),    # the reference's gain, two run() calls
    (TEXTBOOK, 8, 2000, 32, 6, (4, 2)),    # BASELINE configs[4]'s per-GPU share
    (TEXTBOOK, 2, 40, 9, 3, (2, 1)),       # one tile per filter (n = 83), the smallest batch the engine takes (m = 9)
    (TEXTBOOK, 16, 150, 16, 2, (2,)),      # more filters than XCDs
])
def test_batched_instances_equal_solo_lookahead_runs(gpu_required, monkeypatch, quirks, n_inst, N, m, steps, calls):
    # (every run() call but the last has an even number of steps: the single handle pairs updates as they arrive, and the
    # bitwise comparison needs both engines to form the same windows)
    assert all(c % 2 == 0 for c in calls[:-1])
    from conan_slam_amd.synth import Workload

    monkeypatch.setenv("CSLAM_LOOKAHEAD", "1")
    healthy = quirks == REF_EXACT
    loads = [Workload(N, m, np.float32, seed=300 + r, corr=0.02 if healthy else 0.5) for r in range(n_inst)]
    if healthy:
        # the reference's own gain (slam.h:250-260) stays healthy without heading observations on a weakly correlated P0
        # with a small pose block (oracle codes all 0, checked below)
        for w in loads:
            s = np.ones(w.n, np.float32)
            s[:3] = 0.1
            w.P0 = np.asfortranarray(w.P0 * s[:, None] * s[None, :])
    ctrl = [Workload(N, m, np.float32, seed=0, build_p=False).controls(t) for t in range(steps)]
    inputs = _inputs(loads, steps)
    states, traces, flags, nwin = _batch(loads, quirks, ctrl, inputs, steps, calls)
    assert flags == [0] * n_inst, flags
    assert nwin == sum((c + 1) // 2 for c in calls)
    for i, w in enumerate(loads):
        Xs, Ps, trs, fls = _solo(w, quirks, ctrl, inputs[i][0], inputs[i][1], steps)
        assert fls == 0
        Xb, Pb = states[i]
        assert np.array_equal(Xb, Xs), f"instance {i}: state differs from the solo run"
        assert np.array_equal(Pb, Ps), f"instance {i}: covariance differs from the solo run"
        assert traces[i] == trs
    # independent filters: different seeds, different answers
    assert not np.array_equal(states[0][0], states[1][0])
    # ... and the right ones: instance 0 against the oracle (f64 oracle as the fairness reference)
    w = loads[0]
    ref = {}
    for dt in (np.float32, np.float64):
        o = Oracle(dt, quirks)
        X, P = w.X0.astype(dt), np.array(w.P0, dtype=dt, order="F")
        codes = []
        for t in range(steps):
            v, swa = ctrl[t]
            Z, idf = inputs[0][2][t]
            o.predict(X, P, w.n, v, swa, w.QE.astype(dt), w.wb, w.dt)
            codes.append(o.update(X, P, w.n, Z.astype(dt), w.RE.astype(dt), idf, True, fast=True))
        assert codes == [0] * steps, codes
        ref[dt] = (X, P)
    assert_close("batch X", states[0][0], ref[np.float32][0], 1e-5, ref[np.float64][0])
    assert_close("batch P", states[0][1], ref[np.float32][1], 1e-4, ref[np.float64][1])


def test_batch_rejects_what_it_does_not_cover(gpu_required):
    from conan_slam_amd import CslamError, EKFBatch

    with pytest.raises(CslamError):
        EKFBatch(0, 100)
    b = EKFBatch(2, 100)
    with pytest.raises(ValueError):
        b.set_state(0, np.zeros(5, np.float32), np.zeros((5, 5), np.float32))
    with pytest.raises(CslamError):   # m outside 9..32
        b.run(2, np.zeros(2), np.zeros(2), np.eye(2), 1.0, 0.1, [1, 1], [1, 1], 4, np.eye(2))
    with pytest.raises(CslamError):   # an instance without inputs
        b.run(2, np.zeros(2), np.zeros(2), np.eye(2), 1.0, 0.1, [0, 0], [0, 0], 16, np.eye(2))
    b.close()


def test_batch_flags_a_bad_feature_index_per_instance(gpu_required):
    """idf is checked on the device: the instance with the bad index raises CSLAM_FACTOR_BAD_IDF, the other one not."""
    import torch

    from conan_slam_amd import EKFBatch, _capi
    from conan_slam_amd.synth import Workload

    N, m = 200, 16
    loads = [Workload(N, m, np.float32, seed=400 + r) for r in range(2)]
    ctrl = [Workload(N, m, np.float32, seed=0, build_p=False).controls(t) for t in range(2)]
    inputs = _inputs(loads, 2)
    bad = inputs[1][1].clone()
    bad[1, 3] = N + 7
    torch.cuda.synchronize()
    b = EKFBatch(2, N, quirks=TEXTBOOK)
    for i, w in enumerate(loads):
        b.set_state(i, w.X0, w.P0)
    w0 = loads[0]
    b.run(2, [c[0] for c in ctrl], [c[1] for c in ctrl], w0.QE, w0.wb, w0.dt, [inputs[0][0].data_ptr(), inputs[1][0].data_ptr()],
          [inputs[0][1].data_ptr(), bad.data_ptr()], m, w0.RE)
    fl = b.factor_status()
    b.close()
    assert fl[0] == 0 and (fl[1] & _capi.FACTOR_BAD_IDF), fl
