#!/usr/bin/env python3
"""bench.py -- EKF update steps/sec on the MI355X engine (BASELINE.json metric), one JSON line.

Workload at N = 1 (default): BASELINE.json configs[2], the configuration the north-star quotes its
target on -- EKF-SLAM, 5 000 synthetic landmarks (n = 10 003, P = 400 MB), fp32, one MI355X, m = 32
observations per batch update (k = 64).  A "step" is one predict() + one batch update() of the filter
(EKF.cpp:406-455, 481-496) on synthetic inputs (conan_slam_amd/synth.py) that are already
resident in HBM when the timed region starts; nothing returns to the host inside the timed region.

  --workload ekf  (default) the headline line above; sub-records: `per_call` (median wall time of a call with X
                  returned to the host each call), `dropin` (what INTEGRATION.md's adapter does: sync mode,
                  host Z / idf, X read back after predict and after update), `reference_loop` (the cadence of the
                  reference's driver, test/main.cpp:132-200: 6 x (predict + observeHeading) + update + augment).
  --workload mc   BASELINE configs[4]: independent Monte-Carlo EKF instances x 2 000 landmarks, 8 per GPU, each on
                  its own stream pair and host thread (cslam_ekf_run_many).
  --workload pf   BASELINE configs[3]: FastSLAM-2, 512 particles x 1 000 features sharded over the ranks.

Multi-GPU: `--gpus N` starts N ranks itself (torch.distributed.run as a child process, before anything touches
the GPU) unless it already runs under a launcher (RANK / WORLD_SIZE set).  A single EKF does not shard (one dense
P, DESIGN.md "replicas only"): every rank runs independent filter instances with their own seeds, no data-path
collective; value = (sum of steps over ranks) / (max time over ranks); scaling = weak.

The JSON line also carries
  roofline      the covariance downdate kernel (P -= W1 W1^T, slam.h:260), timed live with HIP events on the stream
                it is launched on; `achieved` prices a launch on the bytes the symmetric block-lower algorithm must
                move, the full-storage figure of SURVEY 8d is a separate, clearly named field;
  cpu_baseline  the CPU oracle's dense-order port of the same step (oracle/slam_oracle_fast.c) on one core, plus a
                courtesy all-cores numpy/OpenBLAS row; this host, bounded sample, rank 0 / N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# dense matrix-core peaks the P-GEMMs are priced against.  f32: the datasheet's 157.3 TF (v_mfma_f32_32x32x2_f32 measured
# back to back on the box: 152.7 TF).  f64: the MEASURED rate of the instruction the f64 kernels use, v_mfma_f64_16x16x4_f64
# -- 46.0 TF on all 256 compute units (tools/probes/mfma_peak_probe.hip, profiles/r03_mfma_peak.txt); the datasheet's
# 78.6 TF is not reachable with it (SURVEY 8d asks for the box's own figure where they differ).
MFMA_PEAK_TF = {"f32": 157.3, "f64": 46.0}
MFMA_DATASHEET_TF = {"f32": 157.3, "f64": 78.6}
MFMA_MEASURED_TF = {"f32": 152.7, "f64": 46.0}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=["ekf", "mc", "pf"], default="ekf")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1500,
                    help="untimed steps before the timed region.  The default is long on purpose: the asynchronous loop lets "
                         "the host run hundreds of steps ahead of the GPU, and the first time the backlog passes ~2-3 thousand "
                         "launches the HIP runtime stalls the enqueueing thread once for 30-60 ms while the GPU idles "
                         "(tools/drift_probe.py; f64 N=1000: warm-up 20/400/1000 + 1000 steps -> 80 / 123 / 68.7 us per "
                         "step).  A long warm-up gets that one-time event out of the way")
    ap.add_argument("--preheat-ms", type=float, default=400.0,
                    help="ekf: when --warmup is short (< 400 steps), the same step loop is first run untimed for this long "
                         "so that the W warm-up + K timed steps meet a GPU at its working clocks (a cold MI355X runs the "
                         "first milliseconds ~12 %% slower: 20 steps after 5 measured 8 830 steps/s against 9 950 in a long "
                         "run); the step count is reported as preheat_steps.  0 disables")
    ap.add_argument("--landmarks", type=int, default=None, help="default 5000 (ekf) / 2000 (mc)")
    ap.add_argument("--obs", type=int, default=32, help="observations per batch update (k = 2*obs)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--quirks", choices=["textbook", "ref_exact"], default="textbook",
                    help="gain algebra; identical cost. REF_EXACT on this synthetic map turns every update after the "
                         "first into the reference's LLT-failure no-op (DESIGN.md), so the timed loop uses TEXTBOOK")
    ap.add_argument("--sequential", action="store_true", help="batch=false (EKF.cpp:457-479)")
    ap.add_argument("--defer", type=int, default=-1,
                    help="cslam_ekf_set_deferred: pending W1 columns applied by one P-GEMM (0 = every update at once; "
                         "default: 128 for batches of 32 observations -- one P-GEMM per two updates --, else 0)")
    ap.add_argument("--instances", type=int, default=8, help="mc: filter instances per GPU")
    ap.add_argument("--mc-lanes", type=int, default=1, help="mc, batched engine: split the instances into this many batches")
    ap.add_argument("--mc-engine", choices=["auto", "batch", "handles"], default="auto",
                    help="mc: batched engine (cslam_ekf_batch_*; f32, 9 <= m <= 32) or one handle + host thread per run")
    ap.add_argument("--pgemm-wgs", type=int, default=-1,
                    help="mc: cap on each instance's persistent P-GEMM grid (-1 / 0: whole chip)")
    ap.add_argument("--particles", type=int, default=512)
    ap.add_argument("--features", type=int, default=1000)
    ap.add_argument("--pf-obs", type=int, default=8)
    ap.add_argument("--force-resample", action="store_true", help="pf: resample on every step (worst case exchange)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per_call / dropin / reference_loop sub-records")
    ap.add_argument("--stage-profile", action="store_true", help="extra untimed pass with events around every stage")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="ekf: CSLAM_LOOKAHEAD=0 -- every update runs its own gather / factor / gain chain (round 2's engine)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a box with ONE GPU: every rank uses device 0 and the ranks rendezvous "
                         "over gloo (RCCL refuses one device twice); exercises the launcher / rank logic, not xGMI")
    return ap.parse_args()


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child (nothing has touched the GPU yet)."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        return
    if args.gpus <= 1:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd).returncode)


_REHEARSE = False


def dist_setup(args=None):
    global _REHEARSE
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    _REHEARSE = bool(args is not None and getattr(args, "rehearse_one_gpu", False))
    if _REHEARSE:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if _REHEARSE:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    return rank, local_rank, world, torch, dist


def max_over_ranks(torch, dist, value):
    if dist is None:
        return value
    tt = torch.tensor([value], dtype=torch.float64, device="cpu" if _REHEARSE else "cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline(landmarks, obs, dtype, dname, quirks_name, seconds):
    """The oracle's dense-order port on the SAME workload on this host: one core (the reference is single-threaded),
    plus the courtesy all-cores numpy/OpenBLAS row of SURVEY 8d.  This leg is the only place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from np_restatement import NpSlam            # cpu_baseline leg only
    from pyoracle import Oracle, REF_EXACT, TEXTBOOK  # cpu_baseline leg only

    from conan_slam_amd.synth import Workload

    q = TEXTBOOK if quirks_name == "textbook" else REF_EXACT
    w = Workload(landmarks, obs, dtype, seed=0)
    o = Oracle(dtype, q)
    X, P = w.X0.copy(), w.P0.copy(order="F")
    done, t_total = 0, 0.0
    while True:
        v, swa = w.controls(done)
        Z, idf = w.observations(done)
        t0 = time.perf_counter()
        o.predict(X, P, w.n, v, swa, w.QE, w.wb, w.dt)
        o.update(X, P, w.n, Z, w.RE, idf, True, fast=True)
        dt = time.perf_counter() - t0
        if done > 0:  # the first call pays first-touch of the temporaries
            t_total += dt
        done += 1
        if done >= 2 and (t_total >= seconds or done >= 51):
            break
    timed = done - 1
    out = {
        "value": timed / t_total, "unit": "update steps/s", "cores": 1, "kind": "port",
        "sample": f"{timed} predict+batch-update steps (after 1 untimed) of the same workload: n={w.n}, m={obs}, "
                  f"{dname}, dense operation order of slam.h:235-266 (oracle/slam_oracle_fast.c, gcc -O3 AVX2)",
        "host_cpus": os.cpu_count(),
    }
    # courtesy row: the same dense algebra through numpy / OpenBLAS on all cores (oracle/np_restatement.py)
    try:
        s = NpSlam(dtype, q)
        Xn, Pn = w.X0.copy(), np.array(w.P0, order="F")
        cnt, tt, t = 0, 0.0, done
        budget = max(3.0, seconds / 2)
        while True:
            v, swa = w.controls(t)
            Z, idf = w.observations(t)
            t0 = time.perf_counter()
            Xn, Pn = s.predict(Xn, Pn, v, swa, w.QE, w.wb, w.dt)
            Xn, Pn = s.update(Xn, Pn, Z, w.RE, idf, True)
            dt = time.perf_counter() - t0
            if cnt > 0:
                tt += dt
            cnt += 1
            t += 1
            if cnt >= 3 and (tt >= budget or cnt >= 41):
                break
        out["all_cores_numpy"] = {"value": (cnt - 1) / tt, "unit": "update steps/s", "cores": os.cpu_count(),
                                  "kind": "port (numpy/OpenBLAS dense matrix products, oracle/np_restatement.py)",
                                  "sample": f"{cnt - 1} steps after 1 untimed"}
    except Exception as e:  # the courtesy row must never break the line
        out["all_cores_numpy"] = {"value": None, "error": repr(e)}
    return out


# ------------------------------------------------------------------------------------------------ helpers
class DeviceInputs:
    """Controls and observations of `count` consecutive steps, generated up front and made resident in HBM."""

    def __init__(self, torch, w, count):
        m = w.m
        self.ctrl, self.esize, self.m = [], w.dtype.itemsize, m
        self.Zh = np.zeros((count, 2 * m), dtype=w.dtype)
        self.Ih = np.zeros((count, m), dtype=np.int32)
        self.phi = np.zeros(count)
        for t in range(count):
            self.ctrl.append(w.controls(t))
            Z, idf = w.observations(t)
            self.Zh[t] = Z.reshape(-1, order="F")
            self.Ih[t] = idf
            self.phi[t] = w._true_pose[2]  # (Workload.observations has integrated the true pose up to step t)
        self.dZ = torch.from_numpy(self.Zh).cuda()
        self.dI = torch.from_numpy(self.Ih).cuda()
        torch.cuda.synchronize()
        self.zp, self.ip = self.dZ.data_ptr(), self.dI.data_ptr()

    def z(self, t):
        return self.zp + t * 2 * self.m * self.esize

    def i(self, t):
        return self.ip + t * self.m * 4


def sym_tiles(n):
    t = (n + 127) // 128
    return t * (t + 1) // 2


def pgemm_model(n, k_launch, dname, storage):
    """Bytes one covariance-downdate launch must move and the matrix-core flops it issues."""
    s = 4 if dname == "f32" else 8
    k8 = ((int(round(k_launch)) + 7) // 8) * 8
    full_bytes = 2.0 * n * n * s + 1.0 * n * k_launch * s  # full-storage P read + written, W1 read once
    if storage == "lower" and dname == "f64":
        # ekf_downdate_f64: 128-row tiles on or below the block diagonal are read and written once (the others leave at
        # once), W1 once; passes of 16 columns
        nt = sym_tiles(n)
        depth = (k8 + 15) // 16 * 16
        return {"kernel": "ekf_downdate_f64<2>", "bytes": nt * 128.0 * 128.0 * s * 2 + 1.0 * n * k8 * s,
                "flops_issued": nt * 128.0 * 128.0 * depth * 2, "full_storage_bytes": full_bytes, "n_sym_tiles": nt}
    if storage == "lower":
        nt = sym_tiles(n)
        depth = 64 if k8 <= 64 else (96 if k8 <= 96 else (128 if k8 <= 128 else (k8 + 63) // 64 * 64))  # chunks x chunk depth
        name = ("ekf_downdate_psym4_f32<0,2,32>" if k8 <= 64 else ("ekf_downdate_psym4_f32<0,4,24>" if k8 <= 96 else
                "ekf_downdate_psym4_f32<0,4,32>")) if k8 <= 128 else "ekf_downdate_psym_f32<64,true,false>"
        limbs = int(os.environ.get("CSLAM_PGEMM_LIMBS", "0") or 0)
        kmin = max(57, int(os.environ.get("CSLAM_LIMBS_KMIN", "65") or 65))
        if dname == "f32" and limbs in (6, 9) and kmin <= k8 <= 256:
            # the optional bf16-limb P-GEMM (ekf_pgemm_limbs.hpp): chunks of 16 columns, `limbs` bf16 MFMA products per
            # f32 product; priced against the dense bf16 matrix-core peak
            nch = 4 if k8 <= 64 else (6 if k8 <= 96 else (8 if k8 <= 128 else (12 if k8 <= 192 else 16)))
            return {"kernel": f"ekf_downdate_psym5_bf16<.,{nch},{limbs},3>", "bytes": nt * 65536.0 * 2 + 1.0 * n * k8 * s,
                    "flops_issued": nt * 128.0 * 128.0 * (16 * nch) * 2 * limbs, "full_storage_bytes": full_bytes,
                    "n_sym_tiles": nt, "mfma_peak_tf": 2500.0}
        return {"kernel": name, "bytes": nt * 65536.0 * 2 + 1.0 * n * k8 * s, "flops_issued": nt * 128.0 * 128.0 * depth * 2,
                "full_storage_bytes": full_bytes, "n_sym_tiles": nt}
    tiles = ((n + 127) // 128) ** 2
    return {"kernel": "ekf_downdate_" + dname, "bytes": full_bytes, "flops_issued": tiles * 128.0 * 128.0 * k8 * 2,
            "full_storage_bytes": full_bytes, "n_sym_tiles": None}


def pmc_traffic(kernel, landmarks, k, dname):
    try:
        for e in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]:
            if (e["kernel"], e["landmarks"], e["k"], e["dtype"]) == (kernel, landmarks, int(k), dname):
                return e["traffic_bytes"], e.get("source", "profiles/pmc_traffic.json")
    except Exception:
        pass
    return None, None


def roofline_record(n, k_launch, dname, storage, launch_s, launches, landmarks, extra=None):
    md = pgemm_model(n, k_launch, dname, storage)
    traffic, tsrc = pmc_traffic(md["kernel"], landmarks, k_launch, dname)
    ach = md["bytes"] / launch_s / 1e9 if launch_s and launch_s > 0 else None
    tf = md["flops_issued"] / launch_s / 1e12 if launch_s and launch_s > 0 else None
    # which roof bounds the launch: time to move its bytes at the HBM peak vs time to issue its flops at the MFMA peak
    t_hbm = md["bytes"] / (HBM_PEAK_GBS * 1e9)
    mfma_peak = md.get("mfma_peak_tf", MFMA_PEAK_TF[dname])
    t_mfma = md["flops_issued"] / (mfma_peak * 1e12)
    mfma_bound = t_mfma > t_hbm
    rec = {
        "kernel": md["kernel"], "bound": "mfma" if mfma_bound else "hbm",
        "achieved": tf if mfma_bound else ach, "peak": mfma_peak if mfma_bound else HBM_PEAK_GBS,
        "unit": "TFLOP/s" if mfma_bound else "GB/s",
        "frac": ((tf / mfma_peak) if tf else None) if mfma_bound else ((ach / HBM_PEAK_GBS) if ach else None),
        "hbm_gbs": ach, "hbm_frac": (ach / HBM_PEAK_GBS) if ach else None,
        "roof_times_us": {"hbm": t_hbm * 1e6, "mfma": t_mfma * 1e6},
        "traffic": traffic,
        "traffic_source": (tsrc + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)") if traffic else None,
        "algorithmic_bytes_per_launch": md["bytes"],
        "bytes_model": ("block-lower symmetric storage: n_sym_tiles x (128 x 128 x s bytes) read + written once, W1 panel (n x k8) read once"
                        if storage == "lower" else "full storage: 2 n^2 s + n k s (SURVEY 8d)"),
        "n_sym_tiles": md["n_sym_tiles"],
        "launch_us": launch_s * 1e6 if launch_s else None, "launches_timed": launches, "k_per_launch": k_launch,
        "mfma_tflops_issued": tf, "mfma_frac_of_peak": (tf / mfma_peak) if tf else None,
        "mfma_peaks_tf": {"priced_against": mfma_peak, "datasheet": MFMA_DATASHEET_TF.get(dname),
                          "measured_back_to_back": MFMA_MEASURED_TF.get(dname),
                          "source": "tools/probes/mfma_peak_probe.hip, profiles/r03_mfma_peak.txt"},
        "mfma_flops_model": "issued: tiles x 128^2 x (32-column chunks x 32) x 2 (a symmetric kernel issues ~n^2 k, not 2 n^2 k)",
        # SURVEY 8d's full-storage formula for the same launch: what a non-symmetric implementation would have to
        # move; NOT a fraction of anything this kernel does
        "full_storage_equivalent_gbs": md["full_storage_bytes"] / launch_s / 1e9 if launch_s and launch_s > 0 else None,
    }
    if extra:
        rec.update(extra)
    return rec


def quantiles_ms(ts):
    a = np.sort(np.asarray(ts)) * 1e3
    return {"median_ms": float(np.median(a)), "p10_ms": float(a[int(0.1 * (len(a) - 1))]),
            "p90_ms": float(a[int(0.9 * (len(a) - 1))]), "calls": int(len(a))}


# ------------------------------------------------------------------------------------------------ EKF headline
def ekf_main(args):
    rank, local_rank, world, torch, dist = dist_setup(args)
    dtype = np.float32 if args.dtype == "f32" else np.float64
    N = args.landmarks or 5000

    import conan_slam_amd
    from conan_slam_amd import EKF, Q_REF_EXACT, Q_TEXTBOOK
    from conan_slam_amd.synth import Workload

    if conan_slam_amd.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")

    extras = (world == 1 and not args.no_extras and not args.sequential)
    n_bracket, n_call, n_drop = (64, 64, 48) if extras else (0, 0, 0)
    pre_cap = 8000 if (args.warmup < 400 and args.preheat_ms > 0) else 0  # input slots for the preheat loop
    # short invocations (the driver's `--steps 20 --warmup 5`) are first run COLD -- W warm-up + K timed steps on a GPU
    # that has just been handed to this process -- and reported as `cold_steps_per_sec`; then comes the preheat
    n_cold = (args.warmup + args.steps) if (pre_cap and world == 1) else 0
    pre_cap += n_cold
    total = pre_cap + args.warmup + args.steps
    w = Workload(N, args.obs, dtype, seed=rank)
    n, m, k = w.n, args.obs, 2 * args.obs
    quirks = Q_TEXTBOOK if args.quirks == "textbook" else Q_REF_EXACT
    eng = EKF(N, dtype=dtype, device=local_rank, quirks=quirks, sync_mode=False)
    eng.set_state(w.X0, w.P0)
    if args.defer < 0:
        # the engine's deferred-downdate mode (P = Ps - Wp Wp^T; every update is applied to the state, and to the
        # covariance through the pending-panel correction): one k = 128 P-GEMM per two updates moves P half as often
        # (f64 at N = 1000 gains less from it -- 15 930 vs 15 740 steps/s -- its P-GEMM being latency-bound either way)
        args.defer = 128 if (not args.sequential and 2 * args.obs == 64) else 0
    if args.defer > 0:
        eng.set_deferred(args.defer)
    n_imm = args.steps if (extras and args.defer > 0) else 0
    inp = DeviceInputs(torch, w, total + n_bracket + n_call + n_drop + n_imm)
    batch = not args.sequential

    def step(t):
        v, swa = inp.ctrl[t]
        eng.predict(v, swa, w.QE, w.wb, w.dt)
        eng.update_device(inp.z(t), m, w.RE, inp.i(t), batch=batch)

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    preheat_steps = 0
    cold_rate = None
    if n_cold:
        for t in range(args.warmup):
            step(t)
        barrier()
        c0 = time.perf_counter()
        for t in range(args.warmup, n_cold):
            step(t)
        eng.flush()
        eng.synchronize()
        cold_rate = args.steps / (time.perf_counter() - c0)
        preheat_steps = n_cold
    if pre_cap:
        t_end = time.perf_counter() + args.preheat_ms * 1e-3
        while preheat_steps < pre_cap and time.perf_counter() < t_end:
            for t in range(preheat_steps, min(pre_cap, preheat_steps + 100)):
                step(t)
            preheat_steps = min(pre_cap, preheat_steps + 100)
            eng.synchronize()
        preheat_steps -= n_cold
    for t in range(pre_cap, pre_cap + args.warmup):
        step(t)
    eng.flush()  # (the timed region starts from an applied state: the warm-up's pending P-GEMM is not billed to it,
    barrier()    #  and the K timed steps pay for all of their own, the final flush included)
    # HIP events around a sample of the P-GEMM launches of the timed region, on the stream they run on: one launch in 16
    # (an event pair costs ~11 us of stream time around the kernel it brackets; a short run -- the driver's 20 steps are
    # ten launches -- keeps one in-region sample and is priced on the bracketed pass that follows, see `bracketed_pass`)
    eng.set_profiling(3)
    barrier()
    t0 = time.perf_counter()
    for t in range(pre_cap + args.warmup, total):
        step(t)
    eng.flush()  # the last update's (pending) P-GEMM belongs to the timed region
    eng.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(torch, dist, elapsed)
    stages = eng.stage_times()
    eng.set_profiling(0)
    flags = eng.factor_status()

    if rank != 0:
        eng.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    storage = "full" if (os.environ.get("CSLAM_STORAGE") == "full" or
                         (args.dtype == "f32" and os.environ.get("CSLAM_TUNE_DOWNDATE", "0") not in ("0", "2", "3"))) else "lower"
    dd_ms, dd_cnt = stages["downdate"]
    dd_s = (dd_ms / dd_cnt) * 1e-3 if dd_cnt else None
    # columns one P-GEMM launch applies: k, or (explicit deferral / sequential) the pending columns of several updates
    k_launch = k * max(1, args.defer // k) if args.defer > 0 else k
    t_next = total
    bracket = None
    if extras:
        # every P-GEMM launch bracketed (untimed pass, the filter simply continues): median and spread
        eng.set_profiling(2)
        for t in range(t_next, t_next + n_bracket):
            step(t)
        eng.flush()
        st = eng.stage_times()
        eng.set_profiling(0)
        t_next += n_bracket
        b_ms, b_cnt = st["downdate"]
        bracket = {"launch_us_mean": b_ms / max(b_cnt, 1) * 1e3, "launches": b_cnt}
        if dd_cnt < 16 and b_cnt >= 16:
            # a short timed region samples only a handful of launches (an event pair costs ~11 us of stream time, so
            # not every launch of the timed region is bracketed): the roofline is then priced on the bracketed pass
            # (every launch of 64 further steps), the in-region sample is kept beside it
            bracket["timed_region_sample_us"] = dd_s * 1e6 if dd_s else None
            bracket["timed_region_sample_launches"] = dd_cnt
            dd_s, dd_cnt = (b_ms / b_cnt) * 1e-3, b_cnt
    out = {
        "metric": "ekf_update_steps_per_sec",
        "value": world * args.steps / elapsed,
        "unit": "update steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preheat_steps": preheat_steps,
        "cold_steps_per_sec": cold_rate,
        "value_is": "asynchronous throughput: K steps enqueued back to back, inputs resident in HBM, nothing returned to the "
                    "host inside the timed region; the per-call figure of SURVEY 8d (X returned every call) is "
                    "per_call_steps_per_sec",
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"EKF-SLAM predict+{'batch' if batch else 'sequential'} update, {N} synthetic landmarks "
                        f"(n={n}), m={m} observations/step (k={k}), {args.dtype}, one independent filter per GPU",
            "landmarks": N, "n": n, "obs_per_update": m, "k": k, "gain_algebra": args.quirks,
            "deferred_columns": args.defer,
            "engine": "two-stream pipelined (P-GEMM of update t under the chain of update t+1)"
                      if os.environ.get("CSLAM_PIPELINE", "0") != "0" else
                      ("look-ahead windows: two updates per window, their factor chain (one launch on its own stream) under "
                       "the previous window's P-GEMM, one wide launch per window"
                       if (os.environ.get("CSLAM_LOOKAHEAD", "-1") != "0" and args.dtype == "f32" and n >= 7000
                           and args.defer >= 2 * k and not args.sequential) or os.environ.get("CSLAM_LOOKAHEAD") == "1"
                       else "single stream"),
            "parallelism": f"replicas x{world} (no collective)",
            "baseline_config": "BASELINE.json configs[2]" if (N, args.dtype) == (5000, "f32") else
                               ("BASELINE.json configs[1]" if (N, args.dtype) == (1000, "f64") else "custom"),
        },
        "roofline": roofline_record(n, k_launch, args.dtype, storage, dd_s, dd_cnt, N,
                                    {"bracketed_pass": bracket} if bracket else None),
        "factor_flags": flags,
    }
    if args.stage_profile:
        eng.set_profiling(1)
        for t in range(pre_cap + args.warmup, min(total, pre_cap + args.warmup + 50)):
            step(t)
        st = eng.stage_times()
        eng.set_profiling(0)
        out["stage_us"] = {name: (ms / max(cnt, 1)) * 1e3 for name, (ms, cnt) in st.items()}
    if n_imm:
        # the same K steps with every update's P-GEMM applied at once (cslam_ekf_set_deferred(0)), its own roofline
        eng.flush()
        eng.set_deferred(0)
        for t in range(t_next, t_next + min(10, n_imm)):
            step(t)
        eng.set_profiling(3 if n_imm >= 200 else 4)
        barrier()
        c0 = time.perf_counter()
        for t in range(t_next, t_next + n_imm):
            step(t)
        eng.flush()
        eng.synchronize()
        el = time.perf_counter() - c0
        st = eng.stage_times()
        eng.set_profiling(0)
        t_next += n_imm
        i_ms, i_cnt = st["downdate"]
        out["immediate_mode"] = {
            "value": n_imm / el, "unit": "update steps/s", "ms_per_step": el / n_imm * 1e3,
            "note": "cslam_ekf_set_deferred(0): every update launches its own k = 64 P-GEMM (one sweep of P per update)",
            "roofline": roofline_record(n, k, args.dtype, storage, (i_ms / i_cnt) * 1e-3 if i_cnt else None, i_cnt, N),
        }
        eng.set_deferred(args.defer)
    if extras:
        # per-call times with X returned to the host each call (asynchronous engine, inputs in HBM)
        ts = []
        for t in range(t_next, t_next + n_call):
            c0 = time.perf_counter()
            step(t)
            eng.get_x()
            ts.append(time.perf_counter() - c0)
        t_next += n_call
        out["per_call"] = dict(quantiles_ms(ts[8:]), note="wall time of predict+update with X returned to the host "
                               "after every call (median of the calls after 8 warm-ups); asynchronous mode, Z/idf in HBM",
                               steps_per_sec_at_median=1e3 / quantiles_ms(ts[8:])["median_ms"])
        out["per_call_steps_per_sec"] = out["per_call"]["steps_per_sec_at_median"]  # SURVEY 8d's definition of the metric
        # what INTEGRATION.md's adapter does per call: sync mode (host-side eigen fallback armed), host Z / idf,
        # X read back after predict and after update (slam.h:841-847, 938-943 pass X by reference)
        eng.flush()
        eng.set_deferred(0)  # (a caller that only swaps the class in never asks for deferral)
        eng.set_sync_mode(True)
        ts = []
        for t in range(t_next, t_next + n_drop):
            v, swa = inp.ctrl[t]
            Zt = np.asfortranarray(inp.Zh[t].reshape(2, m, order="F"))
            c0 = time.perf_counter()
            eng.predict(v, swa, w.QE, w.wb, w.dt)
            eng.get_x()
            eng.update(Zt, w.RE, inp.Ih[t], batch=batch)
            eng.get_x()
            ts.append(time.perf_counter() - c0)
        eng.set_sync_mode(False)
        eng.set_deferred(args.defer)
        t_next += n_drop
        q = quantiles_ms(ts[8:])
        out["dropin"] = dict(q, value=1e3 / q["median_ms"], unit="update steps/s",
                             note="drop-in adapter cadence: sync_mode=1, host Z/idf through cslam_ekf_update, "
                                  "cslam_ekf_get_x after predict and after update")
    out["trace_P_end"] = eng.trace()
    out["factor_flags"] = eng.factor_status()
    eng.close()
    if extras and args.dtype == "f32":
        out["reference_loop"] = reference_loop(args, torch, w, N, quirks)
        if quirks != Q_REF_EXACT:
            # the reference-faithful gain (lower-Cholesky factor, n-4 stripe) at size: the heading observation on
            # every control step is what keeps the reference's own filter healthy; factor_flags tells
            # whether it stayed so on this map
            out["reference_loop_ref_exact"] = reference_loop(args, torch, w, N, Q_REF_EXACT)
    if extras:
        out["ref_exact_healthy"] = ref_exact_healthy(args, torch, N, dtype)
    if world == 1 and not args.no_cpu_baseline:
        w.P0 = None
        out["cpu_baseline"] = cpu_baseline(N, args.obs, dtype, args.dtype, args.quirks, args.cpu_baseline_seconds)
        if out["cpu_baseline"]["value"]:
            out["gpu_over_cpu_single_core"] = out["value"] / out["cpu_baseline"]["value"]
        ac = out["cpu_baseline"].get("all_cores_numpy", {}).get("value")
        if ac:
            out["gpu_over_cpu_all_cores_numpy"] = out["value"] / ac
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def ref_exact_healthy(args, torch, N, dtype):
    """One timed number that runs the REFERENCE's own gain (REF_EXACT: lower-Cholesky factor slam.h:250-260, n-4 stripe
    EKF.cpp:442-443) with healthy updates: on SURVEY 8d's strongly correlated P0 that gain turns P indefinite on the first
    update and every later update is the reference's LLT-failure no-op, so this record uses the scenario
    of tests/test_timed_path_gpu.py (b): weakly correlated P0 (U entries N(0, 0.1^2)) and observeHeading on every control
    step as the reference's driver does (test/main.cpp:165-168).  Same kernels as the headline; factor_flags must be 0."""
    from conan_slam_amd import EKF, Q_REF_EXACT
    from conan_slam_amd.synth import Workload

    n_w, n_t = 10, 60  # (the reference's gain keeps this map healthy for ~100 updates: the oracle's first LLT failure is at step 105)
    m = args.obs
    w = Workload(N, m, dtype, seed=0, corr=0.1)
    eng = EKF(N, dtype=dtype, quirks=Q_REF_EXACT, sync_mode=False)
    eng.set_state(w.X0, w.P0)
    w.P0 = None
    if args.defer > 0:
        eng.set_deferred(args.defer)
    inp = DeviceInputs(torch, w, n_w + n_t)

    def step(t):
        v, swa = inp.ctrl[t]
        eng.predict(v, swa, w.QE, w.wb, w.dt)
        eng.observe_heading(float(inp.phi[t]), True)
        eng.update_device(inp.z(t), m, w.RE, inp.i(t), batch=True)

    for t in range(n_w):
        step(t)
    eng.flush()
    eng.synchronize()
    t0 = time.perf_counter()
    for t in range(n_w, n_w + n_t):
        step(t)
    eng.flush()
    eng.synchronize()
    el = time.perf_counter() - t0
    rec = {"value": n_t / el, "unit": "update steps/s", "ms_per_step": el / n_t * 1e3, "steps": n_t,
           "factor_flags": eng.factor_status(), "trace_P_end": eng.trace(), "gain_algebra": "ref_exact",
           "deferred_columns": args.defer,
           "note": "step = predict + observeHeading + batch update (m observations), REF_EXACT quirks, P0 = I + U U^T with U "
                   "entries N(0, 0.1^2); the heading column joins the pending store, so a 128-column window holds one update "
                   "(64 + 1 columns) and every update's P-GEMM runs with k = 65 or 66 (four chunks of 24)"}
    eng.close()
    return rec


def reference_loop(args, torch, w, N, quirks):
    """The cadence of the reference's driver (test/main.cpp:132-200): predict + observeHeading on every control step,
    update (m observations) + augment (1 new landmark) on every 6th."""
    from conan_slam_amd import EKF
    from conan_slam_amd.synth import Workload

    cycles_w, cycles = 10, 150
    w2 = Workload(N, args.obs, w.dtype, seed=0)
    m = args.obs
    eng = EKF(N + cycles_w + cycles + 2, dtype=w.dtype, quirks=quirks, sync_mode=False)
    eng.set_state(w2.X0, w2.P0)
    w2.P0 = None
    steps = []
    for c in range(cycles_w + cycles):
        ctl = []
        for s_ in range(6):
            t = 6 * c + s_
            v, swa = w2.controls(t)
            pose = w2.true_pose_after(t)
            ctl.append((v, swa, float(pose[2])))
        w2._t_obs = None
        # observations from the pose after the 6th control step (Workload.observations advances the true pose itself
        # only when asked in order: it has been advanced above)
        Z, idf = _observe(w2, 6 * c + 5)
        Zn = np.array([[500.0 + c], [0.3]], dtype=w.dtype)
        steps.append((ctl, Z, idf, Zn))
    dZ = torch.from_numpy(np.stack([s[1].reshape(-1, order="F") for s in steps])).cuda()
    dI = torch.from_numpy(np.stack([s[2] for s in steps])).cuda()
    torch.cuda.synchronize()
    es = w.dtype.itemsize

    def cycle(c):
        ctl, _, _, Zn = steps[c]
        for v, swa, phi in ctl:
            eng.predict(v, swa, w2.QE, w2.wb, w2.dt)
            eng.observe_heading(phi, True)
        eng.update_device(dZ.data_ptr() + c * 2 * m * es, m, w2.RE, dI.data_ptr() + c * m * 4, batch=True)
        eng.augment(Zn, w2.RE)

    for c in range(cycles_w):
        cycle(c)
    eng.synchronize()
    t0 = time.perf_counter()
    for c in range(cycles_w, cycles_w + cycles):
        cycle(c)
    eng.flush()
    eng.synchronize()
    el = time.perf_counter() - t0
    rec = {"value": cycles / el, "unit": "observation cycles/s (= update steps/s)", "ms_per_cycle": el / cycles * 1e3,
           "control_steps_per_sec": 6 * cycles / el, "cycles": cycles, "factor_flags": eng.factor_status(),
           "note": "per cycle: 6 x (predict + observeHeading) + batch update (m observations) + augment (1 landmark), "
                   "test/main.cpp:165-189; heading = rank-1 pending column, O(n)"}
    eng.close()
    return rec


def _observe(w, t):
    """Workload.observations for a step whose true pose has already been integrated (reference_loop)."""
    from conan_slam_amd.synth import normal, splitmix64

    N, m = w.N, w.m
    s = 1000 * w.seed
    keys = splitmix64(s + 4 + 7919 * (t + 1), np.arange(N, dtype=np.uint64))
    pick = np.sort(np.argpartition(keys, m - 1)[:m]) if m < N else np.arange(N)
    pose = w._true_pose
    dx, dy = w.LM[0, pick] - pose[0], w.LM[1, pick] - pose[1]
    nz = normal(s + 5 + 104729 * (t + 1), np.arange(2 * m, dtype=np.uint64))
    Z = np.empty((2, m))
    Z[0] = np.sqrt(dx * dx + dy * dy) + nz[0::2] * np.sqrt(float(w.R[0, 0]))
    Z[1] = np.arctan2(dy, dx) - pose[2] + nz[1::2] * np.sqrt(float(w.R[1, 1]))
    return np.asfortranarray(Z.astype(w.dtype)), (pick + 1).astype(np.int32)


# ------------------------------------------------------------------------------------------------ Monte-Carlo
def mc_main(args):
    """BASELINE configs[4]: independent Monte-Carlo EKF runs x 2 000 landmarks, `--instances` per GPU, one stream pair
    and one host thread per run (cslam_ekf_run_many); aggregate update steps/s."""
    rank, local_rank, world, torch, dist = dist_setup(args)
    import ctypes as C

    import conan_slam_amd
    from conan_slam_amd import EKF, Q_REF_EXACT, Q_TEXTBOOK, _capi
    from conan_slam_amd.synth import Workload

    if conan_slam_amd.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    dtype = np.float32 if args.dtype == "f32" else np.float64
    N, m, I = args.landmarks or 2000, args.obs, args.instances
    quirks = Q_TEXTBOOK if args.quirks == "textbook" else Q_REF_EXACT
    total = args.warmup + args.steps
    L = _capi.lib()
    if args.mc_engine == "batch" or (args.mc_engine == "auto" and args.dtype == "f32" and 9 <= m <= 32 and I >= 2):
        return mc_batched(args, rank, local_rank, world, torch, dist, N, m, I, quirks)
    engs, dZs, dIs, n = [], [], [], 3 + 2 * N
    for i in range(I):
        w = Workload(N, m, dtype, seed=100 + rank * I + i)
        e = EKF(N, dtype=dtype, device=local_rank, quirks=quirks, sync_mode=False)
        e.set_state(w.X0, w.P0)
        # (whole-chip P-GEMM grids per instance: capping them at 512 / I was measured slower -- 25.9 k vs 35 - 38 k
        # aggregate steps/s, DESIGN.md 6 -- so the cap is opt-in)
        e.set_pgemm_workgroups(0 if args.pgemm_wgs < 0 else args.pgemm_wgs)
        if args.defer < 0:
            # as the headline: one k = 128 P-GEMM per two updates -- 3.5 launches per step instead of 4, and the runs are
            # launch-bound (measured on one GPU: 34.8 k / 38.4 k / 33.3 k aggregate steps/s with windows of 0 / 128 / 256)
            args.defer = 128 if (args.dtype == "f32" and 2 * m == 64) else 0
        if args.defer > 0:
            e.set_deferred(args.defer)
        w.P0 = None
        inp = DeviceInputs(torch, w, 2 * total)
        engs.append(e)
        dZs.append(inp)
    w0 = Workload(N, m, dtype, seed=0, build_p=False)
    ctrl = [w0.controls(t) for t in range(2 * total)]
    vs = (C.c_double * (2 * total))(*[c[0] for c in ctrl])
    sw = (C.c_double * (2 * total))(*[c[1] for c in ctrl])
    QE = np.asfortranarray(w0.QE)
    RE = np.asfortranarray(w0.RE)

    def run(handles_idx, t0, count):
        cnt = len(handles_idx)
        hs = (C.c_void_p * cnt)(*[engs[i]._h for i in handles_idx])
        zs = (C.c_void_p * cnt)(*[dZs[i].z(t0) for i in handles_idx])
        ids = (C.c_void_p * cnt)(*[dZs[i].i(t0) for i in handles_idx])
        voff = C.cast(C.byref(vs, t0 * 8), C.POINTER(C.c_double))
        soff = C.cast(C.byref(sw, t0 * 8), C.POINTER(C.c_double))
        _capi.check(L.cslam_ekf_run_many(hs, C.c_int(cnt), C.c_int(count), voff, soff, QE.ctypes.data_as(C.c_void_p),
                                         C.c_double(w0.wb), C.c_double(w0.dt), zs, ids, C.c_int(m),
                                         RE.ctypes.data_as(C.c_void_p), C.c_int(1)))

    def sync(idx):
        for i in idx:
            engs[i].flush()
        for i in idx:
            engs[i].synchronize()

    allidx = list(range(I))
    run(allidx, 0, args.warmup)
    sync(allidx)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    for e in engs:
        e.set_profiling(3)
    t0 = time.perf_counter()
    run(allidx, args.warmup, args.steps)
    sync(allidx)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(torch, dist, elapsed)
    dd = [e.stage_times()["downdate"] for e in engs]
    for e in engs:
        e.set_profiling(0)
    flags = [e.factor_status() for e in engs]
    if rank != 0:
        for e in engs:
            e.close()
        if dist is not None:
            dist.destroy_process_group()
        return
    # the same number of steps on ONE instance alone with the whole chip (the filter simply continues)
    engs[0].set_pgemm_workgroups(0)
    run([0], total, args.warmup)
    sync([0])
    t1 = time.perf_counter()
    run([0], total + args.warmup, args.steps)
    sync([0])
    el1 = time.perf_counter() - t1
    dd_ms = sum(d[0] for d in dd)
    dd_cnt = sum(d[1] for d in dd)
    dd_s = dd_ms / dd_cnt * 1e-3 if dd_cnt else None
    storage = "lower" if os.environ.get("CSLAM_STORAGE") != "full" else "full"
    out = {
        "metric": "ekf_update_steps_per_sec", "value": world * I * args.steps / elapsed, "unit": "update steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{world * I} independent Monte-Carlo EKF-SLAM runs x {N} landmarks (n={n}), m={m} (k={2 * m}), "
                               f"{args.dtype}, {I} runs per GPU, one stream pair + one host thread per run",
                   "landmarks": N, "n": n, "obs_per_update": m, "instances_per_gpu": I, "gain_algebra": args.quirks,
                   "deferred_columns": args.defer,
                   "parallelism": f"{I} instances/GPU x {world} GPU(s), no collective",
                   "baseline_config": "BASELINE.json configs[4]" if (N, I * max(world, 1)) == (2000, 64) or N == 2000 else "custom"},
        "single_instance": {"value": args.steps / el1, "unit": "update steps/s", "ms_per_step": el1 / args.steps * 1e3,
                            "note": "one of the instances run alone on the same GPU (same driver)"},
        "concurrency_gain": (I * args.steps / elapsed) / (args.steps / el1) if world == 1 else None,
        "roofline": roofline_record(n, (2 * m) * max(1, args.defer // (2 * m)) if args.defer > 0 else 2 * m, args.dtype, storage,
                                    dd_s, dd_cnt, N,
                                    {"note": "P-GEMM launches of all instances (sampled 1 in 16), co-running with the "
                                             "other instances' kernels"}),
        "factor_flags": flags,
    }
    for e in engs:
        e.close()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, m, dtype, args.dtype, args.quirks, args.cpu_baseline_seconds)
        out["cpu_baseline"]["note"] = "one instance on one core; the reference would run the 64 instances one after another"
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def mc_batched(args, rank, local_rank, world, torch, dist, N, m, I, quirks):
    """configs[4] through the batched engine (cslam_ekf_batch_*): the I runs of a GPU advance in lockstep, every stage
    one launch for all of them; then ONE run alone through a single handle for the concurrency gain."""
    import ctypes as C

    from conan_slam_amd import EKF, EKFBatch, _capi
    from conan_slam_amd.synth import Workload

    dtype, n = np.float32, 3 + 2 * N
    W, K = args.warmup + (args.warmup & 1), args.steps  # (an even warm-up keeps the timed windows whole)
    total = W + K
    # lanes: the I runs as `lanes` batches of I / lanes on their own streams (their stages interleave on the GPU)
    lanes = max(1, args.mc_lanes)
    if I % lanes:
        raise SystemExit("--mc-lanes must divide --instances")
    per = I // lanes
    bs = [EKFBatch(per, N, device=local_rank, quirks=quirks) for _ in range(lanes)]
    inps, first = [], None
    for i in range(I):
        w = Workload(N, m, dtype, seed=100 + rank * I + i)
        bs[i // per].set_state(i % per, w.X0, w.P0)
        if i == 0:
            first = (w.X0.copy(), w.P0)
        w.P0 = None
        inps.append(DeviceInputs(torch, w, total))
    w0 = Workload(N, m, dtype, seed=0, build_p=False)
    ctrl = [w0.controls(t) for t in range(2 * total)]
    v = np.array([c[0] for c in ctrl], dtype=np.float64)
    sw = np.array([c[1] for c in ctrl], dtype=np.float64)

    def run(t0, count):
        for li, b in enumerate(bs):
            mine = inps[li * per:(li + 1) * per]
            b.run(count, v[t0:], sw[t0:], w0.QE, w0.wb, w0.dt, [x.z(t0) for x in mine], [x.i(t0) for x in mine], m, w0.RE)

    def drain():
        for b in bs:
            b.flush()
        for b in bs:
            b.synchronize()

    run(0, W)
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    for b in bs:
        b.set_profiling(8)
    t0 = time.perf_counter()
    run(W, K)
    drain()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(torch, dist, elapsed)
    dd_ms, dd_cnt, flags = 0.0, 0, []
    for b in bs:
        ms_, c_ = b.pgemm_time()
        dd_ms, dd_cnt = dd_ms + ms_, dd_cnt + c_
        b.set_profiling(0)
        flags += b.factor_status()
    if rank != 0:
        for b in bs:
            b.close()
        if dist is not None:
            dist.destroy_process_group()
        return
    # one run alone with the whole chip, through a single handle (its best mode at this size: deferral window 128)
    L = _capi.lib()
    e = EKF(N, dtype=dtype, device=local_rank, quirks=quirks, sync_mode=False)
    e.set_state(*first)
    e.set_deferred(128 if 2 * m == 64 else 0)
    vs = (C.c_double * (2 * total))(*v)
    ss = (C.c_double * (2 * total))(*sw)
    QE, RE = np.asfortranarray(w0.QE), np.asfortranarray(w0.RE)

    def run1(t0, count):
        hs = (C.c_void_p * 1)(e._h)
        zs = (C.c_void_p * 1)(inps[0].z(t0))
        ids = (C.c_void_p * 1)(inps[0].i(t0))
        _capi.check(L.cslam_ekf_run_many(hs, C.c_int(1), C.c_int(count), C.cast(C.byref(vs, t0 * 8), C.POINTER(C.c_double)),
                                         C.cast(C.byref(ss, t0 * 8), C.POINTER(C.c_double)), QE.ctypes.data_as(C.c_void_p),
                                         C.c_double(w0.wb), C.c_double(w0.dt), zs, ids, C.c_int(m),
                                         RE.ctypes.data_as(C.c_void_p), C.c_int(1)))
        e.flush()
        e.synchronize()

    run1(0, W)
    t1 = time.perf_counter()
    run1(W, K)
    el1 = time.perf_counter() - t1
    e.close()
    k_launch = 4 * m  # two updates' panels per launch
    dd_s = dd_ms / dd_cnt * 1e-3 if dd_cnt else None
    rec = roofline_record(n, k_launch, "f32", "lower", dd_s / per if dd_s else None, dd_cnt, N,
                          {"note": f"ONE launch applies the pending panels of all {per} instances of a batch (ekf_downdate_psym4_f32, BATCH "
                                   "mode, tickets over the union of their tiles); bytes, flops and roof times are the launch's, "
                                   f"i.e. {per} x one instance's"})
    rec["launch_us"] = dd_s * 1e6 if dd_s else None
    rec["algorithmic_bytes_per_launch"] *= per
    rec["n_sym_tiles"] *= per
    rec["roof_times_us"] = {k_: t_ * per for k_, t_ in rec["roof_times_us"].items()}
    rec["traffic"], tsrc = pmc_traffic(f"ekf_downdate_psym4_f32<0,4,32> batch x{per}", N, k_launch, "f32")
    rec["traffic_source"] = (tsrc + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)") if tsrc else None
    out = {
        "metric": "ekf_update_steps_per_sec", "value": world * I * K / elapsed, "unit": "update steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{world * I} independent Monte-Carlo EKF-SLAM runs x {N} landmarks (n={n}), m={m} (k={2 * m}), "
                               f"f32, {I} runs per GPU in lockstep through the batched engine (one launch per stage)",
                   "landmarks": N, "n": n, "obs_per_update": m, "instances_per_gpu": I, "batches_per_gpu": lanes,
                   "gain_algebra": args.quirks,
                   "engine": "cslam_ekf_batch (look-ahead windows of two updates, one launch per stage for all instances)",
                   "parallelism": f"{I} instances/GPU x {world} GPU(s), no collective",
                   "baseline_config": "BASELINE.json configs[4]" if N == 2000 else "custom"},
        "single_instance": {"value": K / el1, "unit": "update steps/s", "ms_per_step": el1 / K * 1e3,
                            "note": "instance 0 run alone on the same GPU through a single handle (cslam_ekf_run_many, "
                                    "deferral window 128)"},
        "concurrency_gain": (I * K / elapsed) / (K / el1) if world == 1 else None,
        "roofline": rec,
        "factor_flags": flags,
    }
    for b in bs:
        b.close()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, m, dtype, "f32", args.quirks, args.cpu_baseline_seconds)
        out["cpu_baseline"]["note"] = "one instance on one core; the reference would run the 64 instances one after another"
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ FastSLAM-2
def pf_cpu_baseline(Np, Nf, m, seconds):
    """The oracle's per-particle FastSLAM-2 step (PF.cpp:419-471, 502-544, 222-277) + resample on one core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyoracle import Oracle  # cpu_baseline leg only

    from conan_slam_amd.synth import Workload, normal

    dtype = np.float32
    w = Workload(Nf, m, dtype, seed=0, build_p=False)
    o = Oracle(dtype)
    XF0 = np.asfortranarray(np.stack([w.X0[3::2], w.X0[4::2]]).astype(dtype))
    PF0 = np.asfortranarray(np.tile(np.array([1, 0, 0, 1], dtype=dtype)[:, None], (1, Nf)))
    parts = []
    for g in range(Np):
        pose = np.array([0.05 * normal(77, 3 * g), 0.05 * normal(77, 3 * g + 1), 0.002 * normal(77, 3 * g + 2)], dtype=dtype)
        parts.append([np.array([1.0 / Np], dtype=dtype), pose, np.asfortranarray(np.diag([0.05, 0.05, 1e-4]).astype(dtype)),
                      XF0.copy(order="F"), PF0.copy(order="F")])
    done, tt = 0, 0.0
    while True:
        v, swa = w.controls(done)
        Z, idf = w.observations(done)
        nrm = normal(500 + done, np.arange(3 * Np, dtype=np.uint64)).reshape(3, Np).astype(dtype)
        t0 = time.perf_counter()
        for i, p in enumerate(parts):
            o.pf_predict(p[1], p[2], v, swa, w.QE, w.wb, w.dt)
            o.pf_sample_proposal(p[0], p[1], p[2], p[3], p[4], Z, idf, w.RE, np.ascontiguousarray(nrm[:, i]))
            o.pf_feature_update(p[1], p[3], p[4], Z, idf, w.RE)
        wv = np.array([p[0][0] for p in parts], dtype=dtype)
        sel = ((np.arange(Np) + 0.5) / Np).astype(dtype)
        _, did, keep = o.pf_normalize_resample(wv, Np + 1, True, sel)
        new = [[np.array([wv[i]], dtype=dtype)] + [a.copy(order="F") for a in parts[keep[i]][1:]] for i in range(Np)]
        parts = new
        dt = time.perf_counter() - t0
        if done > 0:
            tt += dt
        done += 1
        if done >= 2 and (tt >= seconds or done >= 21):
            break
    return {"value": (done - 1) / tt, "unit": "PF observation steps/s", "cores": 1, "kind": "port",
            "sample": f"{done - 1} steps (after 1 untimed): {Np} particles x {Nf} features, m={m}, per-particle oracle calls "
                      "(oracle/slam_oracle_pf.inc) + resample with deep copies (PF.cpp:492-498), forced every step",
            "host_cpus": os.cpu_count()}


def pf_main(args):
    """BASELINE configs[3]: Np particles x Nf features, particles block-partitioned over the ranks (strong scaling:
    the particle count is fixed).  A step = predict + sampleProposal + featureUpdate + resampleParticles
    (PF.cpp:419-471, 502-544, 222-277, 473-500); the resample collectives run over torch.distributed (RCCL)."""
    rank, local_rank, world, torch, dist = dist_setup(args)
    from conan_slam_amd.pf import ParticleShard, SingleComm, TorchComm, resample_particles
    from conan_slam_amd.synth import Workload, normal, uniform01

    dtype = np.float32
    Np, Nf, m = args.particles, args.features, args.pf_obs
    assert Np % world == 0, "particles must divide evenly over the ranks"
    L = Np // world
    w = Workload(Nf, m, dtype, seed=0, build_p=False)
    sh = ParticleShard(L, Nf, dtype=dtype, device=local_rank, n_global=Np)
    # every particle: map estimate = truth + N(0,1), PF = I, pose = origin with a small covariance
    XF = np.asfortranarray(np.stack([w.X0[3::2], w.X0[4::2]]).astype(dtype))
    PF = np.asfortranarray(np.tile(np.array([1, 0, 0, 1], dtype=dtype)[:, None], (1, Nf)))
    Pv = np.diag([0.05, 0.05, 1e-4]).astype(dtype)
    for i in range(L):
        g = rank * L + i
        pose = np.array([0.05 * normal(77, 3 * g), 0.05 * normal(77, 3 * g + 1), 0.002 * normal(77, 3 * g + 2)], dtype=dtype)
        sh.set_particle(i, 1.0 / Np, pose, Pv, XF, PF)
    from conan_slam_amd.pf import RcclComm

    if world > 1 and _REHEARSE:
        comm = TorchComm()  # one-GPU rehearsal: RCCL refuses one device twice, the planner runs over gloo instead
    else:
        comm = RcclComm(local_rank) if world > 1 else SingleComm()
    total = args.warmup + args.steps
    inputs = []
    for t in range(total):
        Z, idf = w.observations(t)
        nrm = normal(500 + t, np.arange(3 * Np, dtype=np.uint64)).reshape(3, Np)[:, rank * L:(rank + 1) * L].astype(dtype)
        inputs.append((w.controls(t), Z, idf, np.ascontiguousarray(nrm), uniform01(900 + t, np.arange(Np, dtype=np.uint64))))
    n_resampled = 0

    from conan_slam_amd.pf import stratified_random

    n_eff_min = Np + 1 if args.force_resample else int(0.75 * Np)

    def step(t):
        nonlocal n_resampled
        (v, swa), Z, idf, nrm, u = inputs[t]
        if world == 1:
            # one C call, one staged copy, nothing returned (the resample decision stays on the device)
            sh.observation_step(v, swa, w.QE, w.wb, w.dt, Z, idf, w.RE, nrm, stratified_random(Np, u, dtype), n_eff_min, True)
            return
        sh.predict(v, swa, w.QE, w.wb, w.dt)
        sh.sample_proposal(Z, idf, w.RE, nrm)
        sh.feature_update(Z, idf, w.RE)
        neff, did = resample_particles(sh, comm, n_eff_min, True, uniforms=u)
        n_resampled += int(did)

    for t in range(args.warmup):
        step(t)
    sh.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    n_resampled = 0
    r0 = sh.resample_stats()[1] if world == 1 else 0
    t0 = time.perf_counter()
    for t in range(args.warmup, total):
        step(t)
    sh.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(torch, dist, elapsed)
    if world == 1:
        n_resampled = sh.resample_stats()[1] - r0
    ws = sh.get_weights()
    if rank == 0:
        rec_bytes = (13 + 6 * Nf) * 4
        # the resample's particle moves are the only HBM-heavy part: every kept record is read once and written once
        # into the twin store (pf_gather_move_kernel; single GPU) -- 2 x record x Np bytes per step, resampling or not
        # (an identity copy otherwise); the sharded path packs / exchanges / unpacks only when it resamples
        moved = 2.0 * rec_bytes * Np * (args.steps if world == 1 else n_resampled)
        out = {
            "metric": "pf_observation_steps_per_sec", "value": args.steps / elapsed, "unit": "PF observation steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FastSLAM-2 observation step, {Np} particles x {Nf} features, m={m} observations, "
                                   f"particles sharded {L}/GPU over {world} GPU(s)",
                       "particles": Np, "features": Nf, "obs_per_step": m, "resamples": n_resampled,
                       "force_resample": bool(args.force_resample),
                       "parallelism": f"particles/{world}; RCCL all-reduce(2) + all-gather(N) + grouped send/recv(records) behind the C ABI",
                       "baseline_config": "BASELINE.json configs[3]"},
            "particle_obs_per_sec": args.steps * Np * m / elapsed,
            "weights_finite": bool(np.all(np.isfinite(ws))),
            "roofline": {"kernel": "pf_gather_move_kernel (resample moves)", "bound": "hbm", "unit": "GB/s",
                         "peak": HBM_PEAK_GBS, "achieved": moved / elapsed / 1e9,
                         "frac": moved / elapsed / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "whole-step average: algorithmic bytes of the particle moves (2 x record x Np per step) "
                                 "/ step time; the step is a latency chain of small kernels, not bandwidth-bound "
                                 "(DESIGN.md 5)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = pf_cpu_baseline(Np, Nf, m, args.cpu_baseline_seconds)
            out["gpu_over_cpu_single_core"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    sh.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.no_lookahead:
        os.environ["CSLAM_LOOKAHEAD"] = "0"
    launch_ranks_if_needed(args)
    if args.workload == "pf":
        return pf_main(args)
    if args.workload == "mc":
        return mc_main(args)
    return ekf_main(args)


if __name__ == "__main__":
    main()
