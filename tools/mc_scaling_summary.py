#!/usr/bin/env python3
"""Condenses gpurun_out/prof_mc_scale (tools/mc_scaling.sh) into a table: kernel times of the batched Monte-Carlo engine by
number of instances (rocprofv3 averages, under the profiler) and the un-profiled bench values."""
import csv
import glob
import json
import os
import sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_mc_scale"
print("instances | steps/s (profiled run) | x single | P-GEMM us | wide us | blocks us | rows us | chain us")
for I in (1, 2, 4, 8, 16):
    f = glob.glob(os.path.join(base, f"i{I}", "**", "run_kernel_stats.csv"), recursive=True)
    j = os.path.join(base, f"mc_{I}.json")
    if not f or not os.path.exists(j):
        continue
    d = json.loads(open(j).read().strip().split("\n")[-1])
    t = {}
    for r in csv.DictReader(open(f[0])):
        n = r["Name"]
        if "psym4" in n and "true>" in n:
            t["pgemm"] = float(r["AverageNs"]) / 1e3
        for key, pat in (("wide", "la_wide_batch"), ("blocks", "la_blocks_batch"), ("rows", "la_rows_batch"),
                         ("chain", "la_chain_batch")):
            if pat in n:
                t[key] = float(r["AverageNs"]) / 1e3
    print(f"{I:9d} | {d['value']:10.0f} | {d['concurrency_gain']:.2f} | {t.get('pgemm', 0):7.1f} | {t.get('wide', 0):6.1f} | "
          f"{t.get('blocks', 0):6.1f} | {t.get('rows', 0):5.1f} | {t.get('chain', 0):6.1f}")
