#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection csv (hot-path dispatches only: after the first residual_kernel)."""
import csv, sys
from collections import defaultdict
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import short
rows = []
for r in csv.DictReader(open(sys.argv[1], newline="")):
    rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
first = min((d for d, k, *_ in rows if k.startswith("residual")), default=0)
agg = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for d, k, c, v, t in rows:
    if d >= first and not k.startswith(("at::", "__amd")):
        agg[k][c].append(v); dur[k].append(t)
names = sorted({c for k in agg for c in agg[k]})
print(f"{'kernel':42s} {'n':>5s} {'us':>8s} " + " ".join(f"{c[-16:]:>16s}" for c in names))
for k in sorted(agg, key=lambda k: -sum(dur[k])):
    n = len(next(iter(agg[k].values())))
    print(f"{k[:42]:42s} {n:5d} {sum(dur[k]) / len(dur[k]) / 1e3:8.1f} " + " ".join(f"{sum(agg[k][c]) / max(1, len(agg[k][c])):16.0f}" for c in names))
