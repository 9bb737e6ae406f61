#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- the fixtures that pin the C oracle.

The values come from the INDEPENDENT numpy/scipy restatement (oracle/np_restatement.py: matrix-level numpy
+ LAPACK), not from the C oracle they are used to check.  Neither can be compared with the reference itself
(it is not buildable here and ships no vectors: "parity unpinned"), so the fixtures pin the
two restatements against each other and freeze today's behaviour against regressions.

Run in the build container only:  python tools/gen_golden.py      (rewrites tests/golden/)
Fixtures are inputs + expected outputs of each hot-path function at n in {3, 5, 13, 53} and k in {0, 2, 6, 14},
in f32 and f64, REF_EXACT and TEXTBOOK, plus the summary of the bundled 30-landmark demo run.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from np_restatement import NpSlam, REF_EXACT, TEXTBOOK  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def scenario(n, dtype, seed):
    rng = np.random.default_rng(seed)
    U = rng.normal(size=(n, 3)) * 0.3
    P = np.eye(n) + U @ U.T
    s = np.ones(n)
    s[:3] = 0.1
    P = P * s[:, None] * s[None, :]
    X = np.concatenate([[0.5, -1.0, 0.2], rng.uniform(-400, 400, size=n - 3)])
    return X.astype(dtype), P.astype(dtype)


def observations(X, idf, dtype, seed):
    rng = np.random.default_rng(seed)
    X = X.astype(np.float64)
    Z = np.zeros((2, len(idf)))
    for i, f in enumerate(idf):
        fx = 3 + 2 * (f - 1)
        dx, dy = X[fx] - X[0], X[fx + 1] - X[1]
        Z[0, i] = np.hypot(dx, dy) + rng.normal() * 0.3
        Z[1, i] = np.arctan2(dy, dx) - X[2] + rng.normal() * 0.02
    return Z.astype(dtype)


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {}
    Q = np.diag([0.18, 6.0e-4])
    R = np.diag([0.08, 2.4e-3])
    for dname, dtype in (("f32", np.float32), ("f64", np.float64)):
        for qname, quirks in (("ref", REF_EXACT), ("txt", TEXTBOOK)):
            s = NpSlam(dtype, quirks)
            Qd, Rd = Q.astype(dtype), R.astype(dtype)
            for n in (3, 5, 13, 53):
                N = (n - 3) // 2
                X, P = scenario(n, dtype, seed=n)
                tag = f"{dname}_{qname}_n{n}"
                # predict
                Xp, Pp = s.predict(X, P, 83.33, 0.07, Qd, 73.0, 0.01)
                cases[f"{tag}_predict"] = dict(X=X, P=P, Q=Qd, v=83.33, swa=0.07, wb=73.0, dt=0.01, Xo=Xp, Po=Pp)
                # heading
                Xh, Ph = s.observe_heading(X, P, 0.27, True)
                cases[f"{tag}_heading"] = dict(X=X, P=P, phi=0.27, Xo=Xh, Po=Ph)
                # augment by two
                Zn = np.array([[150.0, 600.0], [0.5, -1.2]], dtype=dtype)
                Xa, Pa = s.augment(X, P, Zn, Rd)
                cases[f"{tag}_augment"] = dict(X=X, P=P, Z=Zn, R=Rd, Xo=Xa, Po=Pa)
                for k in (0, 2, 6, 14):
                    m = k // 2
                    if m > N:
                        continue
                    idf = (np.random.default_rng(100 + k).permutation(N)[:m] + 1).astype(np.int32)
                    Z = observations(X, idf, dtype, seed=k) if m else np.zeros((2, 0), dtype=dtype)
                    for batch in (True, False):
                        Xu, Pu = s.update(X, P, Z, Rd, idf, batch)
                        cases[f"{tag}_update_k{k}_{'batch' if batch else 'seq'}"] = dict(
                            X=X, P=P, Z=Z, R=Rd, idf=idf, batch=int(batch), Xo=Xu, Po=Pu)
    # particle-filter pieces (f32 REF only + f64), one particle
    for dname, dtype in (("f32", np.float32), ("f64", np.float64)):
        s = NpSlam(dtype, REF_EXACT)
        rng = np.random.default_rng(7)
        Xv = np.array([1.0, -2.0, 0.15], dtype=dtype)
        A = rng.normal(size=(3, 3)) * np.array([0.3, 0.3, 0.02])[:, None]
        Pv = (A @ A.T + np.diag([0.05, 0.05, 1e-4])).astype(dtype)
        XF = rng.uniform(-200, 200, size=(2, 5)).astype(dtype)
        PF = []
        for _ in range(5):
            B = rng.normal(size=(2, 2)) * 0.5
            PF.append((B @ B.T + 0.2 * np.eye(2)).astype(dtype))
        Rd = R.astype(dtype)
        idf = np.array([2, 5, 3], dtype=np.int32)
        Z = np.zeros((2, 3), dtype=dtype)
        for i, f in enumerate(idf):
            dx, dy = float(XF[0, f - 1] - Xv[0]), float(XF[1, f - 1] - Xv[1])
            Z[0, i] = np.hypot(dx, dy) + 0.1 * (i - 1)
            Z[1, i] = np.arctan2(dy, dx) - float(Xv[2]) + 0.004 * (1 - i)
        normals = np.array([0.3, -1.1, 0.6], dtype=dtype)
        wn, Xs, Ps = s.pf_sample_proposal(dtype(0.01), Xv, Pv, XF, PF, Z, idf, Rd, normals)
        XFu, PFu = s.pf_feature_update(Xs, XF, PF, Z, idf, Rd)
        Xpp, Ppp = s.pf_predict(Xv, Pv, 83.33, 0.05, Q.astype(dtype), 73.0, 0.01)
        ge = s.pf_gauss_evaluate(np.array([0.2, -0.01], dtype=dtype), (np.diag([0.3, 0.002]) + 0.001).astype(dtype))
        cases[f"{dname}_pf"] = dict(Xv=Xv, Pv=Pv, XF=XF, PF=np.stack([p.reshape(-1, order="F") for p in PF], axis=1),
                                    Z=Z, idf=idf, R=Rd, normals=normals, w0=dtype(0.01), w=wn, Xs=Xs,
                                    XFu=XFu, PFu=np.stack([p.reshape(-1, order="F") for p in PFu], axis=1),
                                    Xp=Xpp, Pp=Ppp, Q=Q.astype(dtype), gauss=ge)
        wts = rng.uniform(0.01, 1.0, 24).astype(dtype)
        sel = (np.arange(24) + rng.uniform(size=24)) / 24.0
        keep, neff = s.pf_stratified_resample(wts, sel.astype(dtype), ref_exact=False)
        keepr, _ = s.pf_stratified_resample(wts, sel.astype(dtype), ref_exact=True)
        cases[f"{dname}_resample"] = dict(w=wts, select=sel.astype(dtype), keep=keep, keep_ref_exact=keepr, neff=neff)

    flat = {}
    for cname, d in cases.items():
        for key, val in d.items():
            flat[f"{cname}/{key}"] = np.asarray(val)
    np.savez_compressed(os.path.join(OUT, "hotpath_cases.npz"), **flat)
    print(f"wrote {len(cases)} cases, {len(flat)} arrays -> tests/golden/hotpath_cases.npz")

    # config 1: the bundled demo map, noise off, driven through the numpy restatement
    from sim_driver import SlamConfig, load_demo_map, run_demo

    class NpBackend:
        def __init__(self, dtype):
            self.s = NpSlam(dtype, REF_EXACT)
            self.X = np.zeros(3, dtype=dtype)
            self.P = np.zeros((3, 3), dtype=dtype)

        n = property(lambda self: self.X.shape[0])

        def predict(self, v, swa, Q, wb, dt):
            self.X, self.P = self.s.predict(self.X, self.P, v, swa, np.asarray(Q), wb, dt)

        def observe_heading(self, phi, use):
            self.X, self.P = self.s.observe_heading(self.X, self.P, phi, use)

        def update(self, Z, R, idf, batch):
            self.X, self.P = self.s.update(self.X, self.P, np.asarray(Z), np.asarray(R), idf, batch)

        def augment(self, Z, R):
            self.X, self.P = self.s.augment(self.X, self.P, np.asarray(Z), np.asarray(R))

        def get_x(self):
            return self.X.copy()

        def get_p(self):
            return self.P.copy()

    LM, WP = load_demo_map()
    summary = {}
    for dname, dtype in (("f32", np.float32), ("f64", np.float64)):
        for sig in (True, False):
            r = run_demo(NpBackend(dtype), LM, WP, SlamConfig(), int_signum=sig)
            summary[f"{dname}_{'int_signum' if sig else 'plain_sign'}"] = {
                k: r[k] for k in ("steps", "obs_events", "updates", "max_m", "mean_m", "final_n", "trace_P", "X_pose",
                                  "XTrue", "wp_switch_steps")}
            print(dname, sig, summary[f"{dname}_{'int_signum' if sig else 'plain_sign'}"])
    json.dump({"_generated_by": "tools/gen_golden.py (numpy restatement, noise off, REF_EXACT)", **summary},
              open(os.path.join(OUT, "demo_run_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
