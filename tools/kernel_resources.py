#!/usr/bin/env python3
"""Prints VGPR / SGPR / LDS / scratch per kernel from the gfx950 assembly of the engine (hipcc -S)."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = "/tmp/cslam_kernels.s"
    for src in ("cslam_ekf.hip", "cslam_ekf_batch.hip", "cslam_pf.hip"):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "conan_slam_amd", "csrc", src)], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
        for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, flags=re.S):
            g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
            if len(sys.argv) > 1 and sys.argv[1] not in name:
                continue
            print(f"{name:60s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>3s} "
                  f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")


if __name__ == "__main__":
    main()
