#!/bin/bash
# per-launch durations of one kernel (default residual_kernel) under different bench modes; run on the GPU box
# usage: tools/residual_trace.sh [kernel-prefix] -- "<bench args>" ["<bench args>" ...]
cd /tmp && export TMPDIR=/tmp
K=${1:-residual_kernel}; shift; shift
i=0
for args in "$@"; do
  i=$((i+1)); rm -rf /tmp/rt_$i
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rt_$i -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --steps 12 --warmup 2 --profile-scans 0 $args > /tmp/rt_$i.log 2>&1
  echo "== $args"; grep '^{' /tmp/rt_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stages_ms_serial'])"
  grep -i 'error\|Traceback' -A5 /tmp/rt_$i.log | head -20
  f=$(find /tmp/rt_$i -name '*kernel_trace.csv' | head -1)
  python - "$f" "$K" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(sys.argv[2])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; last 32 (us):", " ".join(f"{x:.0f}" for x in d[-32:]))
PY
done
