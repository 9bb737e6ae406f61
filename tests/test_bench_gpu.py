"""bench.py as the driver runs it: a child process, one JSON line with the contract's fields (the workloads at reduced
sizes so that the test stays short; the numbers themselves are not asserted)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _run(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_driver_shape_prints_the_contract_line(gpu_required):
    d = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--landmarks", "1500", "--cpu-baseline-seconds", "2",
              "--preheat-ms", "50"])
    for k in CONTRACT + ["cpu_baseline"]:
        assert k in d, k
    assert d["metric"] == "ekf_update_steps_per_sec" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["value"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and d["factor_flags"] == 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1.0 and r["unit"] in ("GB/s", "TFLOP/s")
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1


def test_bench_batched_monte_carlo_mode(gpu_required):
    d = _run(["--workload", "mc", "--instances", "3", "--landmarks", "400", "--steps", "12", "--warmup", "4", "--no-cpu-baseline"])
    for k in CONTRACT:
        assert k in d, k
    assert d["config"]["instances_per_gpu"] == 3 and "cslam_ekf_batch" in d["config"]["engine"]
    assert d["factor_flags"] == [0, 0, 0] and d["value"] > 0 and d["single_instance"]["value"] > 0
