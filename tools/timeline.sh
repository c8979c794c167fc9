#!/bin/bash
# kernel timeline (start, duration, queue, name) of the last scans of a bench run; run on the GPU box
# usage: tools/timeline.sh <n_rows> <bench args...>
cd /tmp && export TMPDIR=/tmp
N=$1; shift
rm -rf /tmp/tl
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --steps 40 --warmup 5 --profile-scans 0 --extra-configs 0 "$@" > /tmp/tl.log 2>&1
grep '^{' /tmp/tl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['scan_thread_ms'])"
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python - "$f" "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]); rows = rows[-(n + 200):-200]   # a window of the steady state, not the wind-down
t0 = int(rows[0]["Start_Timestamp"])
qs = {}
for r in rows:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{s:9.1f} {d:7.1f} q{q} {'    ' * q}{r['Kernel_Name'][:60]}")
PY
