#!/usr/bin/env python3
"""Dump a bench workload as raw little-endian files for the torch-free harness scripts/micro/pickbench (no GPU needed):
   python scripts/dump_workload.py [--config 5] [--out scripts/micro/_gen/c5]
   meta.txt: R P B n_index index_slots n_scorers kind weight ...   pods.bin [P x 64 B]   index_hashes.bin u64   index_pods.bin u32
   reqs.bin [R x (8 + 8 B)] (batch 0 = the workload's own requests; pickbench derives its other batches from it on the device)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "scripts", "micro", "_gen", "c5"))
    a = ap.parse_args()
    pkg = graft.load_package()
    wl = pkg.workload.make_workload(a.config)
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, "meta.txt"), "w") as f:
        f.write(" ".join(str(x) for x in [wl.R, wl.P, wl.B, len(wl.index_hashes), wl.index_slots, len(wl.chain)] + [int(v) for kw in wl.chain for v in kw]) + "\n")
    wl.pods.tofile(os.path.join(a.out, "pods.bin"))
    wl.index_hashes.astype("<u8").tofile(os.path.join(a.out, "index_hashes.bin"))
    wl.index_pods.astype("<u4").tofile(os.path.join(a.out, "index_pods.bin"))
    wl.reqs.astype("<u8").tofile(os.path.join(a.out, "reqs.bin"))
    print(a.out, {k: os.path.getsize(os.path.join(a.out, k)) for k in sorted(os.listdir(a.out))})


if __name__ == "__main__":
    main()
