#!/usr/bin/env python3
"""bench.py -- EKF update steps/sec on the MI355X engine (BASELINE.json metric), one JSON line.

Workload at N = 1 (default): BASELINE.json configs[2], the configuration the north-star quotes its
target on -- EKF-SLAM, 5 000 synthetic landmarks (n = 10 003, P = 400 MB), fp32, one MI355X, m = 32
observations per batch update (k = 64).  A "step" is one predict() + one batch update() of the filter
(EKF.cpp:406-455, 481-496) on synthetic inputs (conan_slam_amd/synth.py) that are already
resident in HBM when the timed region starts; nothing returns to the host inside the timed region.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): a single EKF does not shard (one dense
P, DESIGN.md "replicas only"), so every rank runs an independent filter instance of the same size with
its own seed (the Monte-Carlo arrangement of BASELINE configs[4]); no data-path collective; the value
is (sum of steps over ranks) / (max time over ranks); scaling = weak.

The JSON line also carries
  roofline      the downdate kernel (P -= W1 W1^T, slam.h:260), timed live with HIP events on the engine's
                stream around every launch of the timed region;
  cpu_baseline  the CPU oracle's dense-order port of the same update (oracle/slam_oracle_fast.c), timed on
                this host on a bounded sample, rank 0 / N = 1 only.  Reported, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TF = {"f32": 157.3, "f64": 78.6}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=["ekf", "pf"], default="ekf",
                    help="ekf: the headline EKF update metric (default). pf: BASELINE configs[3], FastSLAM-2 observation "
                         "steps with the particle set sharded over the ranks and the RCCL resample exchange")
    ap.add_argument("--particles", type=int, default=512)
    ap.add_argument("--features", type=int, default=1000)
    ap.add_argument("--pf-obs", type=int, default=8)
    ap.add_argument("--force-resample", action="store_true", help="pf: resample on every step (worst case exchange)")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--landmarks", type=int, default=5000)
    ap.add_argument("--obs", type=int, default=32, help="observations per batch update (k = 2*obs)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--quirks", choices=["textbook", "ref_exact"], default="textbook",
                    help="gain algebra; identical cost. REF_EXACT on this synthetic map turns every update after the "
                         "first into the reference's LLT-failure no-op (DESIGN.md), so the timed loop uses TEXTBOOK")
    ap.add_argument("--sequential", action="store_true", help="batch=false (EKF.cpp:457-479)")
    ap.add_argument("--defer", type=int, default=0,
                    help="cslam_ekf_set_deferred: pending W1 columns applied by one P-GEMM (0 = every update at once)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-deferred-extra", action="store_true",
                    help="skip the additional deferred-mode measurement reported under 'deferred_mode'")
    ap.add_argument("--stage-profile", action="store_true", help="extra untimed pass with events around every stage")
    return ap.parse_args()


def cpu_baseline(args, dtype):
    """Times the oracle's dense-order port on the SAME workload on this host (1 core)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyoracle import Oracle, REF_EXACT, TEXTBOOK  # cpu_baseline leg only

    from conan_slam_amd.synth import Workload

    w = Workload(args.landmarks, args.obs, dtype, seed=0)
    o = Oracle(dtype, TEXTBOOK if args.quirks == "textbook" else REF_EXACT)
    X, P = w.X0.copy(), w.P0.copy(order="F")
    done, t_total = 0, 0.0
    while done < 2 or (t_total < args.cpu_baseline_seconds and done < 50):
        v, swa = w.controls(done)
        Z, idf = w.observations(done)
        t0 = time.perf_counter()
        o.predict(X, P, w.n, v, swa, w.QE, w.wb, w.dt)
        o.update(X, P, w.n, Z, w.RE, idf, True, fast=True)
        dt = time.perf_counter() - t0
        if done > 0 or args.cpu_baseline_seconds <= 0:  # the first call pays first-touch of the temporaries
            t_total += dt
        done += 1
        if done >= 2 and t_total >= args.cpu_baseline_seconds:
            break
    timed = max(done - 1, 1)
    return {
        "value": timed / t_total if t_total > 0 else None,
        "unit": "update steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{timed} predict+batch-update steps (after 1 untimed) of the same workload: n={w.n}, m={args.obs}, "
                  f"{args.dtype}, dense operation order of slam.h:235-266 (oracle/slam_oracle_fast.c, gcc -O3 AVX2)",
        "host_cpus": os.cpu_count(),
    }


def pf_main(args):
    """BASELINE configs[3]: Np particles x Nf features, particles block-partitioned over the ranks (strong scaling:
    the particle count is fixed).  A step = predict + sampleProposal + featureUpdate + resampleParticles
    (PF.cpp:419-471, 502-544, 222-277, 473-500); the resample collectives run over torch.distributed (RCCL)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
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
    comm = TorchComm(device=torch.device("cuda", local_rank)) if world > 1 else SingleComm()
    total = args.warmup + args.steps
    inputs = []
    for t in range(total):
        Z, idf = w.observations(t)
        nrm = normal(500 + t, np.arange(3 * Np, dtype=np.uint64)).reshape(3, Np)[:, rank * L:(rank + 1) * L].astype(dtype)
        inputs.append((w.controls(t), Z, idf, np.ascontiguousarray(nrm), uniform01(900 + t, np.arange(Np, dtype=np.uint64))))
    n_resampled = 0

    def step(t):
        nonlocal n_resampled
        (v, swa), Z, idf, nrm, u = inputs[t]
        sh.predict(v, swa, w.QE, w.wb, w.dt)
        sh.sample_proposal(Z, idf, w.RE, nrm)
        sh.feature_update(Z, idf, w.RE)
        neff, did = resample_particles(sh, comm, Np + 1 if args.force_resample else int(0.75 * Np), True, uniforms=u)
        n_resampled += int(did)

    for t in range(args.warmup):
        step(t)
    sh.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    n_resampled = 0
    t0 = time.perf_counter()
    for t in range(args.warmup, total):
        step(t)
    sh.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ws = sh.get_weights()
    if rank == 0:
        print(json.dumps({
            "metric": "pf_observation_steps_per_sec", "value": args.steps / elapsed, "unit": "PF observation steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FastSLAM-2 observation step, {Np} particles x {Nf} features, m={m} observations, "
                                   f"particles sharded {L}/GPU over {world} GPU(s)",
                       "particles": Np, "features": Nf, "obs_per_step": m, "resamples": n_resampled,
                       "force_resample": bool(args.force_resample),
                       "parallelism": f"particles/{world}; all-reduce(2) + all-gather(N) + all-to-all-v(records)",
                       "baseline_config": "BASELINE.json configs[3]"},
            "particle_obs_per_sec": args.steps * Np * m / elapsed,
            "weights_finite": bool(np.all(np.isfinite(ws))),
        }), flush=True)
    sh.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.workload == "pf":
        return pf_main(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dtype = np.float32 if args.dtype == "f32" else np.float64
    esize = np.dtype(dtype).itemsize

    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import conan_slam_amd
    from conan_slam_amd import EKF, Q_REF_EXACT, Q_TEXTBOOK
    from conan_slam_amd.synth import Workload

    if conan_slam_amd.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")

    total_steps = args.warmup + args.steps
    # second, clearly labelled measurement on the default line: the same steps with cslam_ekf_set_deferred(128)
    with_deferred_extra = (world == 1 and args.defer == 0 and not args.sequential and args.dtype == "f32"
                           and not args.no_deferred_extra)
    extra_steps = args.steps if with_deferred_extra else 0
    w = Workload(args.landmarks, args.obs, dtype, seed=rank)
    n, m, k = w.n, args.obs, 2 * args.obs
    quirks = Q_TEXTBOOK if args.quirks == "textbook" else Q_REF_EXACT
    eng = EKF(args.landmarks, dtype=dtype, device=local_rank, quirks=quirks, sync_mode=False)
    eng.set_state(w.X0, w.P0)
    if args.defer > 0:
        eng.set_deferred(args.defer)
    w.P0 = None  # free 400 MB of host memory

    # inputs of every step, generated up front and made resident in HBM
    ctrl = []
    Zall = np.zeros((total_steps + extra_steps, 2 * m), dtype=dtype)
    Iall = np.zeros((total_steps + extra_steps, m), dtype=np.int32)
    for t in range(total_steps + extra_steps):
        ctrl.append(w.controls(t))
        Z, idf = w.observations(t)
        Zall[t] = Z.reshape(-1, order="F")
        Iall[t] = idf
    dZ = torch.from_numpy(Zall).cuda()
    dI = torch.from_numpy(Iall).cuda()
    torch.cuda.synchronize()
    zp, ip = dZ.data_ptr(), dI.data_ptr()
    batch = not args.sequential

    def step(t):
        v, swa = ctrl[t]
        eng.predict(v, swa, w.QE, w.wb, w.dt)
        eng.update_device(zp + t * 2 * m * esize, m, w.RE, ip + t * m * 4, batch=batch)

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for t in range(args.warmup):
        step(t)
    barrier()
    # HIP events around one P-GEMM launch in sixteen of the timed region, on the engine's stream (an event pair costs
    # about 11 us of stream time around the kernel it brackets, so bracketing every launch would slow the loop by 9 %)
    eng.set_profiling(3)
    barrier()
    t0 = time.perf_counter()
    for t in range(args.warmup, total_steps):
        step(t)
    eng.flush()  # deferred mode: the last pending panels are applied inside the timed region
    eng.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stages = eng.stage_times()
    eng.set_profiling(0)
    flags = eng.factor_status()
    trace_end = eng.trace()

    stage_profile = None
    if args.stage_profile and rank == 0:
        eng.set_profiling(1)
        for t in range(args.warmup, min(total_steps, args.warmup + 50)):
            step(t)
        st = eng.stage_times()
        eng.set_profiling(0)
        stage_profile = {name: (ms / max(cnt, 1)) * 1e3 for name, (ms, cnt) in st.items()}  # us per launch

    deferred_extra = None
    if with_deferred_extra:
        # the engine's deferred-downdate mode (P = Ps - Wp Wp^T, one P-GEMM per 128 pending columns = 2 steps here):
        # same kernels otherwise, final flush inside the timed region, the filter simply continues
        eng.set_deferred(128)
        eng.set_profiling(3)
        barrier()
        t1 = time.perf_counter()
        for t in range(total_steps, total_steps + extra_steps):
            step(t)
        eng.flush()
        eng.synchronize()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        st2 = eng.stage_times()
        eng.set_profiling(0)
        eng.set_deferred(0)
        d_ms, d_cnt = st2["downdate"]
        deferred_extra = {
            "value": extra_steps / el2, "unit": "update steps/s", "ms_per_step": el2 / extra_steps * 1e3,
            "deferred_columns": 128, "p_gemm_launches": d_cnt, "p_gemm_launch_us": d_ms / max(d_cnt, 1) * 1e3,
            "factor_flags": eng.factor_status(),
            "note": "cslam_ekf_set_deferred(128): every update is applied (state, and covariance through the pending-panel "
                    "correction); the P-GEMM runs once per 128 pending W1 columns; final flush inside the timed region",
        }

    if rank != 0:
        eng.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    dd_ms, dd_cnt = stages["downdate"]
    dd_s = (dd_ms / max(dd_cnt, 1)) * 1e-3
    # columns one P-GEMM launch applies: k, or (deferred / sequential) the pending columns of several updates
    k_launch = (k if batch else 2 * m) * args.steps / max(dd_cnt, 1) if args.defer > 0 else (k if batch else 2 * m)
    # algorithmic bytes of one downdate launch: P read once + written once, W1 read once
    dd_bytes = 2.0 * n * n * esize + 1.0 * n * k_launch * esize
    dd_flops = 2.0 * n * n * k_launch
    achieved = dd_bytes / dd_s / 1e9 if dd_s > 0 else None
    tune_dd = os.environ.get("CSLAM_TUNE_DOWNDATE", "0")
    storage = "full" if (os.environ.get("CSLAM_STORAGE") == "full" or args.dtype == "f64" or
                         tune_dd not in ("0", "2", "3")) else "lower"
    if args.dtype != "f32" or tune_dd in ("1", "4"):
        kernel_name = "ekf_downdate_" + args.dtype
    elif tune_dd == "2" or k_launch > 128 or (k_launch > 64 and storage != "lower"):
        kernel_name = "ekf_downdate_psym_f32<64,true,%s>" % ("false" if storage == "lower" else "true")
    elif storage == "lower" and tune_dd == "0":
        # memory operations inside the MFMA loop (the shipped path): two chunks of 32 for k <= 64, four for k <= 128
        kernel_name = "ekf_downdate_psym4_f32<0,%d>" % (2 if k_launch <= 64 else 4)
    else:
        kernel_name = "ekf_downdate_psym3_f32<%s>" % ("false,false" if storage == "lower" else "true,true")
    traffic = None  # physical HBM bytes per launch: from the committed rocprofv3 --pmc summary of this exact configuration
    try:
        for e in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]:
            if (e["kernel"], e["landmarks"], e["k"], e["dtype"]) == (kernel_name, args.landmarks, int(k_launch), args.dtype):
                traffic = e["traffic_bytes"]
    except Exception:
        traffic = None
    out = {
        "metric": "ekf_update_steps_per_sec",
        "value": world * args.steps / elapsed,
        "unit": "update steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"EKF-SLAM predict+{'batch' if batch else 'sequential'} update, {args.landmarks} synthetic landmarks "
                        f"(n={n}), m={m} observations/step (k={k}), {args.dtype}, one independent filter per GPU",
            "landmarks": args.landmarks,
            "n": n,
            "obs_per_update": m,
            "k": k,
            "gain_algebra": args.quirks,
            "deferred_columns": args.defer,
            "parallelism": f"replicas x{world} (no collective)",
            "baseline_config": "BASELINE.json configs[2]" if (args.landmarks, args.dtype) == (5000, "f32") else "custom",
        },
        "roofline": {
            "kernel": kernel_name,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
            "traffic": traffic,
            "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)" if traffic else None,
            "note": "achieved uses SURVEY 8d's full-storage algorithmic bytes (2 n^2 s + n k s); the symmetric kernel "
                    "physically moves about half of them (traffic), so frac can exceed 1; issued MFMA flops are n^2 k "
                    "(half of the 2 n^2 k the mfma_* fields are normalised by)",
            "algorithmic_bytes_per_launch": dd_bytes,
            # the same launch priced on the bytes it physically moved (PMC traffic): the figure to read for how close
            # the kernel runs to the fabric
            "physical_gbs": (traffic / dd_s / 1e9) if (traffic and dd_s > 0) else None,
            "physical_frac": (traffic / dd_s / 1e9 / HBM_PEAK_GBS) if (traffic and dd_s > 0) else None,
            "launch_us": dd_s * 1e6,
            "launches_timed": dd_cnt,
            "k_per_launch": k_launch,
            "mfma_tflops_issued": dd_flops / dd_s / 1e12 if dd_s > 0 else None,
            "mfma_frac_of_peak": (dd_flops / dd_s / 1e12 / MFMA_PEAK_TF[args.dtype]) if dd_s > 0 else None,
        },
        "factor_flags": flags,
        "deferred_mode": deferred_extra,
        "trace_P_end": trace_end,
    }
    if stage_profile:
        out["stage_us"] = stage_profile
    if world == 1 and not args.no_cpu_baseline:
        eng.close()
        out["cpu_baseline"] = cpu_baseline(args, dtype)
        if out["cpu_baseline"]["value"]:
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    else:
        eng.close()
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
