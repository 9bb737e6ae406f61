"""The reference-side binding EXECUTED: include/cslam_adapter.hpp (HipEKF / HipPF : public EKF / PF) built with g++ on the
GPU box against the Eigen-free stand-in of the reference's types, linked against libcslam_hip.so, and driven the way
the reference's driver drives its back-ends -- through std::shared_ptr<Slam> (test/main.cpp:89, 165-189, 279-311).

The call stream is the bundled 30-landmark demo run (test/main.cpp:24-200; map literal in tests/golden/demo_map.json):
the harness-side simulator (vehicle, steering, sensor, known-association table -- filter-independent with the noise
switches off) runs through the CPU oracle and every filter call it makes (predict, observeHeading, update, augment) is
recorded; tests/adapter/adapter_replay.cpp replays the record through the adapter.  Final n / X / trace(P) against the
oracle on the same stream and against tests/golden/demo_run_summary.json (the independent numpy restatement's run) at
the whole-demo tolerance of SURVEY 8d.  One FastSLAM-2 observation step goes through HipPF the same way.

PARITY UNPINNED.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from pyoracle import Oracle, REF_EXACT

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fmt(a):
    return " ".join("%.9g" % float(x) for x in np.asarray(a, dtype=np.float64).reshape(-1, order="F"))


class _Recorder:
    """A filter back-end for sim_driver.run_demo that forwards to the oracle back-end and writes every call down."""

    def __init__(self, inner, out):
        self.inner, self.out = inner, out

    @property
    def n(self):
        return self.inner.n

    def predict(self, v, swa, Q, wb, dt):
        self.out.write(f"P {_fmt([v, swa, wb, dt])} {_fmt(Q)}\n")
        self.inner.predict(v, swa, Q, wb, dt)

    def observe_heading(self, phi, use):
        self.out.write(f"H {_fmt([phi])} {1 if use else 0}\n")
        self.inner.observe_heading(phi, use)

    def update(self, Z, R, idf, batch):
        m = Z.shape[1] if Z.size else 0
        if m:  # (the reference calls update with an empty Z too: a no-op on both sides, EKF.cpp:101-123)
            self.out.write(f"U {1 if batch else 0} {m} {_fmt(Z)} {' '.join(str(int(i)) for i in idf)} {_fmt(R)}\n")
        return self.inner.update(Z, R, idf, batch)

    def augment(self, Z, R):
        q = Z.shape[1] if Z.size else 0
        if q:
            self.out.write(f"A {q} {_fmt(Z)} {_fmt(R)}\n")
        self.inner.augment(Z, R)

    def get_x(self):
        return self.inner.get_x()

    def get_p(self):
        return self.inner.get_p()


def _build_runner(tmp_path):
    from conan_slam_amd import _capi

    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    exe = str(tmp_path / "adapter_replay")
    libdir = os.path.dirname(os.path.abspath(_capi.LIB_PATH))
    cmd = [gxx, "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "tests", "adapter"), os.path.join(ROOT, "tests", "adapter", "adapter_replay.cpp"),
           "-L" + libdir, "-lcslam_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_demo_run_and_pf_step_through_the_adapter(gpu_required, tmp_path):
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sim_driver import OracleBackend, load_demo_map, run_demo
    from test_pf_gpu import _obs_for, _random_particles

    LM, WP = load_demo_map()
    stream = tmp_path / "demo_stream.txt"
    with open(stream, "w") as out:
        ref = run_demo(_Recorder(OracleBackend(np.float32), out), LM, WP)          # the whole demo, noise off
        # ---- one FastSLAM-2 observation step (main.cpp:279-311) on 64 particles x 6 features
        dtype, npart, nf, m = np.float32, 64, 6, 3
        parts = _random_particles(npart, nf, dtype, seed=404)
        rng = np.random.default_rng(405)
        for p in parts:
            p[0] = dtype(rng.uniform(0.0, 1.0) ** 4)
        out.write(f"F {npart} {nf}\n")
        for w, Xv, Pv, XF, PF in parts:
            out.write(f"{_fmt([w])} {_fmt(Xv)} {_fmt(Pv)} {_fmt(XF)} {_fmt(PF)}\n")
        Q = np.diag([0.18, 6e-4]).astype(dtype)
        R = np.diag([0.08, 0.0024]).astype(dtype)
        idf = np.array([2, 5, 3], dtype=np.int32)
        Z = _obs_for(parts, idf, dtype, seed=9)
        normals = rng.normal(size=(3, npart)).astype(dtype)
        select = ((np.arange(npart) + rng.uniform(size=npart)) / npart).astype(dtype)
        nmin = npart + 1  # forced resample
        out.write(f"p {_fmt([83.33, 0.04, 73.0, 0.01])} {_fmt(Q)}\n")
        out.write(f"s {m} {_fmt(Z)} {' '.join(str(int(i)) for i in idf)} {_fmt(R)} {_fmt(normals)}\n")
        out.write(f"r {nmin} {_fmt(select)}\n")
    hi = run_demo(OracleBackend(np.float64), LM, WP)
    assert ref["final_n"] <= 3 + 2 * 32

    exe = _build_runner(tmp_path)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, str(stream), "32"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    assert len(r.stdout.splitlines()) == 1, "the adapter printed errors: " + r.stdout[:2000]
    got = json.loads(lines[-1])

    # ---- EKF: the whole demo run through std::shared_ptr<Slam> -> HipEKF
    assert got["n"] == ref["final_n"] == 53 and got["factor_flags"] == 0
    Xg, Xo, Xh = np.array(got["X"]), ref["X"].astype(np.float64), hi["X"]
    # whole-demo tolerance of SURVEY 8d (22 015 control steps in f32): the engine must be as close to the f64 run as the
    # f32 CPU oracle is, within a factor; the trace to 1e-2
    e_g, e_o = np.abs(Xg - Xh).max(), np.abs(Xo - Xh).max()
    assert np.abs(Xg - Xo).max() <= 1e-4 * np.abs(Xo).max() or e_g <= 8.0 * e_o + 1e-4 * np.abs(Xo).max(), (e_g, e_o)
    assert abs(got["trace_P"] - ref["trace_P"]) <= 1e-2 * abs(ref["trace_P"]), (got["trace_P"], ref["trace_P"])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "demo_run_summary.json")))["f32_int_signum"]
    assert got["n"] == gold["final_n"]
    assert abs(got["trace_P"] - gold["trace_P"]) <= 1e-2 * gold["trace_P"], (got["trace_P"], gold["trace_P"])
    assert np.abs(Xg[:2] - np.array(gold["X_pose"][:2])).max() <= 0.1 and abs(Xg[2] - gold["X_pose"][2]) <= 1e-3

    # ---- PF: predict + sampleProposal + featureUpdate + resampleParticles through HipPF
    o = Oracle(np.float32, REF_EXACT)
    for i, p in enumerate(parts):
        o.pf_predict(p[1], p[2], 83.33, 0.04, Q, 73.0, 0.01)
        w = np.array([p[0]], dtype=dtype)
        o.pf_sample_proposal(w, p[1], p[2], p[3], p[4], Z, idf, R, normals[:, i].copy())
        p[0] = w[0]
        o.pf_feature_update(p[1], p[3], p[4], Z, idf, R)
    wv = np.array([p[0] for p in parts], dtype=dtype)
    neff, did, keep = o.pf_normalize_resample(wv, nmin, True, select)
    assert did and got["pf"]["resampled"] == 1
    assert abs(got["pf"]["neff"] - float(neff)) <= 1e-2 * float(neff)
    G = np.array(got["pf"]["particles"])
    assert G.shape == (npart, 4 + 2 * nf)
    assert np.allclose(G[:, 0], 1.0 / npart, rtol=1e-6)
    for i in range(npart):
        src = parts[keep[i]]
        exp = np.concatenate([src[1], src[3].reshape(-1, order="F")]).astype(np.float64)
        assert np.abs(G[i, 1:] - exp).max() <= 2e-5 * max(1.0, np.abs(exp).max()), (i, int(keep[i]))
