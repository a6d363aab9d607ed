#!/usr/bin/env python3
"""Per-kernel VGPRs / LDS / occupancy of one csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles,
no GPU needed).  python tools/kernel_usage.py lmaze_foveal.hip [name-fragment]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def usage(src):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                          "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "gym-lmaze_amd", "csrc", src),
                          "-o", "/dev/null"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return kernels


if __name__ == "__main__":
    frag = sys.argv[2] if len(sys.argv) > 2 else ""
    for k, v in sorted(usage(sys.argv[1]).items()):
        if frag in k:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print("%-90s %s" % (name.split("(")[0][-90:], v))
