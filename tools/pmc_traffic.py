#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, csv) into HBM bytes per launch per kernel.

Units and gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: the counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes read, so the read side is doubled:  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
Only dispatches after the first residual_kernel launch are used (the per-scan hot path; the map pre-build of bench.py is excluded).

The output carries `_meta`: a fingerprint of the kernel sources the passes ran on (bench.kernel_sources_sha) and the commit -- bench.py refuses the file
when its own sources differ.

usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json
"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    n = name.strip('"')
    if n.startswith("void "):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):   # cut the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i]
    return n


def load(path, counter):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    first = next((d for d, k, _ in rows if k.startswith("residual")), 0)
    agg = defaultdict(list)
    for d, k, v in rows:
        if d >= first:
            agg[k].append(v)
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
        out[k] = {"launches": len(fetch.get(k, [])), "fetch_kib_per_launch": round(f, 3), "write_kib_per_launch": round(w, 3),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        from bench import kernel_sources_sha
        sha = kernel_sources_sha()
    except Exception:   # noqa: BLE001
        sha = None
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("IMMESH_COMMIT")
    except Exception:   # noqa: BLE001
        commit = os.environ.get("IMMESH_COMMIT")
    out["_meta"] = {"kernel_sources_sha16": sha, "commit": commit}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k[:70]:70s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch'] / 1e6:9.3f} MB/launch")


if __name__ == "__main__":
    main()
