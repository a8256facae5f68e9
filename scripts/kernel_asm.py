#!/usr/bin/env python3
"""Compile one translation unit of csrc/ with -save-temps and report, for the kernels whose mangled name contains a pattern,
VGPR / SGPR / spill counts and the instruction mix of every loop (by nesting depth).  A development aid for the instruction
diet of the pick kernels:   python scripts/kernel_asm.py eppk_pick_quad.hip 'pick_quad_kernelImLb1ELb0'"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gateway-api-inference-extension_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]


def cls(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"


def main():
    unit, pat = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    tmp = "/tmp/kernel_asm"
    os.makedirs(tmp, exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-c", "-o", os.path.join(tmp, "u.o"), os.path.join(CSRC, unit), "-save-temps=obj"],
                   check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    sfile = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(tmp, sfile)).read()
    for m in re.finditer(r"^(_Z\w+):", s, re.M):
        name = m.group(1)
        if pat not in name: continue
        body = s[m.end():s.index(".Lfunc_end", m.end())]
        meta = s[s.index(".name:           " + name):][:900]
        res = dict(re.findall(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\d+)", meta))
        print(name[:110]); print("  ", res)
        open(os.path.join(tmp, "kernel.s"), "w").write(body)
        tot = {}
        for line in body.split("\n"):
            t = line.strip()
            if not t or t.startswith(";") or t.startswith("."): continue
            c = cls(t.split()[0]); tot[c] = tot.get(c, 0) + 1
        print("   whole function:", tot)


if __name__ == "__main__":
    main()
