#!/usr/bin/env python3
"""Static instruction counts per source line of one kernel's ISA (hipcc -S -gline-tables-only): which lines of the hot path
cost how many VALU / SALU / memory instructions.  `python scripts/isa_by_line.py file.s [first_line last_line]`."""
import collections
import re
import sys

path = sys.argv[1]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
files = {}
cur = None
cnt = collections.defaultdict(lambda: collections.Counter())
for raw in open(path):
    t = raw.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', t)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|buffer_[a-z0-9_]+|global_[a-z0-9_]+|scratch_[a-z0-9_]+|flat_[a-z0-9_]+)\b", t)
    if not m or cur is None:
        continue
    op = m.group(1)
    if op.startswith("v_"):
        k = "valu"
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
        k = "wait"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        k = "br"
    elif op.startswith("s_load") or op.startswith("s_buffer"):
        k = "smem"
    elif op.startswith("s_"):
        k = "salu"
    elif op.startswith("ds_"):
        k = "lds"
    else:
        k = "vmem"
    cnt[cur][k] += 1
tot = collections.Counter()
for (f, l) in sorted(cnt):
    if "kernels" not in f or not (lo <= l <= hi):
        continue
    c = cnt[(f, l)]
    tot.update(c)
    print(f"{l:5d}  valu {c['valu']:4d} salu {c['salu']:4d} br {c['br']:3d} vmem {c['vmem']:3d} lds {c['lds']:3d} smem {c['smem']:3d} wait {c['wait']:3d}")
print("total", dict(tot))
