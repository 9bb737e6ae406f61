#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: for a window of dispatches in the steady state, each kernel's start and
end relative to the first one, its stream/queue, and how much of it overlapped the covariance downdate (P-GEMM).
Usage: python tools/trace_timeline.py <kernel_trace.csv> [skip_fraction=0.5] [count=40]"""
import csv
import sys


def short(name):
    n = name.split("(")[0]
    for pre in ("void cslam::", "cslam::", "void "):
        if n.startswith(pre):
            n = n[len(pre):]
    return n[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i0 = int(len(rows) * skip)
    # start the window at a P-GEMM
    while i0 < len(rows) and "downdate" not in rows[i0]["Kernel_Name"]:
        i0 += 1
    win = rows[i0:i0 + count]
    if not win:
        print("no rows")
        return
    t0 = int(win[0]["Start_Timestamp"])
    dd = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win if "downdate" in r["Kernel_Name"]]
    qkey = "Queue_Id" if "Queue_Id" in win[0] else ("Stream_Id" if "Stream_Id" in win[0] else None)
    print(f"{'kernel':44s} {'queue':>6s} {'start_us':>9s} {'end_us':>9s} {'dur_us':>8s} {'under P-GEMM':>12s}")
    for r in win:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ov = 0
        if "downdate" not in r["Kernel_Name"]:
            for a, b in dd:
                ov += max(0, min(e, b) - max(s, a))
        q = r.get(qkey, "") if qkey else ""
        print(f"{short(r['Kernel_Name']):44s} {q:>6s} {(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} "
              f"{(100.0 * ov / max(e - s, 1)):11.0f}%")
    if len(dd) >= 2:
        per = (dd[-1][0] - dd[0][0]) / (len(dd) - 1) / 1e3
        busy = sum(b - a for a, b in dd[:-1]) / (len(dd) - 1) / 1e3
        print(f"P-GEMM period {per:.1f} us, P-GEMM busy {busy:.1f} us per period ({100 * busy / per:.0f} %)")


if __name__ == "__main__":
    main()
