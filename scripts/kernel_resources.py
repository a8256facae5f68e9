#!/usr/bin/env python3
"""Register / LDS / spill figures of the pick-kernel instantiations, from hipcc's -Rpass-analysis=kernel-resource-usage
(CPU only: cross-compiles for gfx950).  `python scripts/kernel_resources.py [u64_6] [filter-substring]`."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gateway-api-inference-extension_amd", "csrc")
unit = sys.argv[1] if len(sys.argv) > 1 else "u64_6"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", "-o", "/tmp/_kr.o",
       f"eppk_pick_{unit}.hip", "-Rpass-analysis=kernel-resource-usage", *extra]
out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void eppk::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k:110s} VGPR {v.get('VGPRs')} SGPR {v.get('TotalSGPRs')} occ {v.get('Occupancy [waves/SIMD]')} scratch {v.get('ScratchSize [bytes/lane]')} "
              f"spillS {v.get('SGPRs Spill')} spillV {v.get('VGPRs Spill')}")
