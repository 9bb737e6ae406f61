#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (counter_collection.csv + kernel_trace.csv) per kernel: mean counter value and
mean duration per dispatch.  Usage: python tools/pmc_summary.py gpurun_out/pmc [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    base = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for d in sorted(glob.glob(os.path.join(base, "*", "*")) + glob.glob(os.path.join(base, "*"))):
        if not os.path.isdir(d):
            continue
        cc = glob.glob(os.path.join(d, "*_counter_collection.csv"))
        kt = glob.glob(os.path.join(d, "*_kernel_trace.csv"))
        if not cc:
            continue
        dur = defaultdict(list)
        if kt:
            for r in csv.DictReader(open(kt[0])):
                dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        vals = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(cc[0])):
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", os.path.relpath(d, base))
        for kname, cs in vals.items():
            if want not in kname:
                continue
            short = kname.split("(")[0][-60:]
            dd = dur.get(kname, [0])
            msg = f"  {short:62s} n={len(dd):3d} dur_us={sum(dd)/max(len(dd),1)/1e3:9.1f}"
            for c, v in cs.items():
                msg += f"  {c}={sum(v)/len(v):.4g}"
            print(msg)


if __name__ == "__main__":
    main()
