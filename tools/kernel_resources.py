#!/usr/bin/env python3
"""Per-kernel register / spill / LDS / occupancy report of a translation unit (device-only compile, no GPU needed).
usage: tools/kernel_resources.py [scldm_amd/csrc/api.hip] [filter-regex]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tu = sys.argv[1] if len(sys.argv) > 1 else "scldm_amd/csrc/api.hip"
flt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", *os.environ.get("SCLDM_HIPCC_FLAGS", "").split(),
       "--cuda-device-only", "-c", tu, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=root, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.+?): (\S+)\s*\[", line) or re.search(r"remark:\s+(.+?): (\S+)\s*\[", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if not flt.search(name):
        continue
    g = lambda key: v.get(key, "?")
    print(f"{name[:100]:100s} vgpr {g('VGPRs'):>4} agpr {g('AGPRs'):>4} spill {g('VGPRs Spill'):>4} scratch {g('ScratchSize [bytes/lane]'):>5} "
          f"lds {g('LDS Size [bytes/block]'):>6} occ {g('Occupancy [waves/SIMD]')}")
