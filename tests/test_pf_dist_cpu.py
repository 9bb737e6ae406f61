"""CPU tests of the multi-shard resample logic (PF.cpp:473-500 over a block-partitioned particle set):
exchange planning, and a world_size-2 run over torch.distributed `gloo` with a numpy stand-in for the GPU shard.
The stand-in only stores records; selection arithmetic is the product code in conan_slam_amd/pf.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conan_slam_amd.pf import SingleComm, plan_exchange, resample_particles, stratified_keep, stratified_random
from pyoracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyShard:
    """Test double with ParticleShard's resample surface; a record is [w, payload...]."""

    def __init__(self, weights, payload, n_global):
        self.dtype = np.dtype(weights.dtype)
        self.w = weights.copy()
        self.payload = payload.copy()  # (n_local, L)
        self.n_local = weights.shape[0]
        self.n_global = n_global

    def weight_sums(self):
        w = self.w.astype(np.float64)
        return float(w.sum()), float((w * w).sum())

    def scale_weights(self, s):
        self.w = (self.w * self.dtype.type(s)).astype(self.dtype)

    def set_uniform_weight(self, w0):
        self.w[:] = self.dtype.type(w0)

    def get_weights(self):
        return self.w.copy()

    def weights_tensor(self):
        import torch

        return torch.from_numpy(self.w.copy())

    def pack(self, idx):
        import torch

        rec = np.concatenate([self.w[idx, None], self.payload[idx]], axis=1) if len(idx) else \
            np.zeros((0, 1 + self.payload.shape[1]), self.dtype)
        return torch.from_numpy(np.ascontiguousarray(rec))

    def unpack(self, idx, buf):
        rec = buf.numpy()
        for j, d in enumerate(idx):
            self.w[d] = rec[j, 0]
            self.payload[d] = rec[j, 1:]

    def gather_local(self, keep, w_new):
        self.payload = self.payload[keep].copy()
        self.w[:] = self.dtype.type(w_new)


def _weights(n, seed, skew):
    return (np.random.default_rng(seed).uniform(0.0, 1.0, n) ** skew).astype(np.float32)


def test_stratified_keep_matches_the_oracle():
    o = Oracle(np.float32)
    for n, seed in ((8, 1), (64, 2), (200, 3)):
        w = _weights(n, seed, 5)
        u = np.random.default_rng(seed + 10).uniform(size=n)
        sel = stratified_random(n, u)
        assert np.allclose(sel, o.pf_stratified_random(n, u.astype(np.float32)), rtol=1e-6)
        wn = (w / w.sum(dtype=np.float32)).astype(np.float32)
        keep_ref, _, _ = o.pf_stratified_resample(w, sel, ref_exact=False)
        assert np.array_equal(stratified_keep(wn, sel), keep_ref)
        assert np.all(np.diff(keep_ref) >= 0) and keep_ref.min() >= 0 and keep_ref.max() < n


def test_plan_exchange_is_consistent_across_ranks():
    n_local, world = 5, 4
    n = n_local * world
    keep = np.sort(np.random.default_rng(0).integers(0, n, n)).astype(np.int32)
    plans = [plan_exchange(keep, r, world, n_local) for r in range(world)]
    for s in range(world):
        for d in range(world):
            assert plans[s][1][d] == plans[d][3][s]  # what s sends to d is what d expects from s
    assert sum(sum(p[1]) for p in plans) == n and all(sum(p[3]) == n_local for p in plans)
    # simulate the exchange and compare with the direct gather
    ids = np.arange(n)
    out = np.full(n, -1)
    for d in range(world):
        off = {s: 0 for s in range(world)}
        pos = 0
        for s in range(world):
            cnt = plans[d][3][s]
            send_src, send_counts = plans[s][0], plans[s][1]
            start = sum(send_counts[:d])
            chunk = send_src[start:start + cnt]
            for j in range(cnt):
                out[d * n_local + plans[d][2][pos + j]] = ids[s * n_local + chunk[j]]
            pos += cnt
    assert np.array_equal(out, ids[keep])


@pytest.mark.parametrize("skew,expect", [(6, True), (0.05, False)])
def test_single_shard_resample_matches_the_oracle(skew, expect):
    o = Oracle(np.float32)
    n = 48
    w = _weights(n, 5, skew)
    payload = np.arange(n, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32)
    sh = NumpyShard(w, payload, n)
    u = np.random.default_rng(6).uniform(size=n)
    sel = stratified_random(n, u)
    neff, did = resample_particles(sh, SingleComm(), int(0.75 * n), True, select=sel)
    wref = w.copy()
    neff_ref, did_ref, keep = o.pf_normalize_resample(wref, int(0.75 * n), True, sel)
    assert did == did_ref == expect and abs(neff - float(neff_ref)) < 1e-3 * float(neff_ref)
    if expect:
        assert np.array_equal(sh.payload[:, 0].astype(int), keep) and np.allclose(sh.w, 1.0 / n)
    else:
        assert np.allclose(sh.w, wref, rtol=1e-6) and np.array_equal(sh.payload, payload)
    # resampling switched off: only the normalisation happens (PF.cpp:490)
    sh2 = NumpyShard(w, payload, n)
    _, did2 = resample_particles(sh2, SingleComm(), n, False, select=sel)
    assert not did2 and abs(float(sh2.w.sum()) - 1.0) < 1e-5


_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["CSLAM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["CSLAM_ROOT"], "tests"))
sys.path.insert(0, os.path.join(os.environ["CSLAM_ROOT"], "oracle"))
import torch, torch.distributed as dist
from conan_slam_amd.pf import TorchComm, resample_particles, stratified_random
from test_pf_dist_cpu import NumpyShard, _weights
dist.init_process_group("gloo")
comm = TorchComm()
n_local, n = 12, 12 * comm.world
skew = float(os.environ["CSLAM_SKEW"])
w_all = _weights(n, 9, skew)
payload_all = np.stack([np.arange(n), 100 + np.arange(n)], axis=1).astype(np.float32)
lo = comm.rank * n_local
sh = NumpyShard(w_all[lo:lo + n_local], payload_all[lo:lo + n_local], n)
sel = stratified_random(n, np.random.default_rng(10).uniform(size=n))
neff, did = resample_particles(sh, comm, int(0.75 * n), True, select=sel)
out = [None] * comm.world
dist.all_gather_object(out, {"w": sh.w.tolist(), "p": sh.payload.tolist(), "neff": neff, "did": did})
if comm.rank == 0:
    print("RESULT" + json.dumps(out))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("skew,expect", [(6, True), (0.05, False)])
def test_two_rank_resample_over_gloo(tmp_path, skew, expect):
    """world_size 2, gloo: all-reduce of [sum w, sum w^2], all-gather of weights, all-to-all-v of records."""
    import json

    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, CSLAM_ROOT=ROOT, CSLAM_SKEW=str(skew), OMP_NUM_THREADS="1", CSLAM_HIP_RUNTIME="system")
    port = 29600 + (os.getpid() % 300) + (1 if expect else 0)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0]
    out = json.loads(line[len("RESULT"):])
    n = 24
    w_all = _weights(n, 9, skew)
    o = Oracle(np.float32)
    sel = stratified_random(n, np.random.default_rng(10).uniform(size=n))
    wref = w_all.copy()
    neff_ref, did_ref, keep = o.pf_normalize_resample(wref, int(0.75 * n), True, sel)
    assert out[0]["did"] == out[1]["did"] == did_ref == expect
    assert abs(out[0]["neff"] - float(neff_ref)) < 1e-3 * float(neff_ref)
    got_ids = np.array(out[0]["p"] + out[1]["p"])[:, 0].astype(int)
    got_w = np.array(out[0]["w"] + out[1]["w"])
    if expect:
        assert np.array_equal(got_ids, keep) and np.allclose(got_w, 1.0 / n)
    else:
        assert np.array_equal(got_ids, np.arange(n)) and np.allclose(got_w, wref, rtol=1e-6)


def test_vectorised_strata_positions_equal_the_reference_loop():
    """stratified_random is a running sum in the particle dtype (PF.cpp:579-596): the vectorised form must reproduce
    the loop's roundings bit for bit."""
    from conan_slam_amd.pf import stratified_random

    def loop(n, uniforms, dtype):
        t = np.dtype(dtype).type
        k = t(1) / t(n)
        out = np.empty(n, dtype=dtype)
        di = k / t(2)
        for i in range(n):
            if i > 0:
                di = t(di + k)
            out[i] = t(di + (t(uniforms[i]) * k - k / t(2)))
        return out

    rng = np.random.default_rng(0)
    for dt in (np.float32, np.float64):
        for n in (1, 2, 7, 64, 512, 4097):
            u = rng.uniform(size=n)
            assert np.array_equal(loop(n, u, dt), stratified_random(n, u, dt))
