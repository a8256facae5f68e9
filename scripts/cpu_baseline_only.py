#!/usr/bin/env python3
"""bench.py's cpu_baseline leg alone (no GPU, no torch): the C5 workload, 16 rotating batches, both CPU algorithms.
Used to read the CPU figures of a host without paying for a bench run:  python scripts/cpu_baseline_only.py [--config 5] [--batches 16]"""
import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--batches", type=int, default=16)
    a = ap.parse_args()
    pkg = graft.load_package()
    orc = graft.load_oracle()
    orc.build()
    wl = pkg.workload.make_workload(a.config)
    batches = bench.make_batches(pkg, wl, types.SimpleNamespace(config=a.config), a.batches)
    cb, _, _ = bench.cpu_baseline(wl, orc, batches[-1], batches)
    cb["host_threads"] = os.cpu_count()
    print(json.dumps(cb))


if __name__ == "__main__":
    main()
