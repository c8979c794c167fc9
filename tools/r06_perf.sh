#!/bin/bash
# round 5: the quick perf check of a change: the headline line (20 scans), the 500-scan steady state and a steady-state timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; T=${1:-perf}
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>$O/${T}_20.err | grep '^{' | tail -1 > $O/${T}_20.json
timeout 300 python $R/bench.py --gpus 1 --gpu-scans 1 --steps 500 --warmup 20 --nu-scans 0 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>$O/${T}_500.err | grep '^{' | tail -1 > $O/${T}_500.json
python - $O/${T}_20.json $O/${T}_500.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"], d.get("scan_thread_ms"))
PY
bash $R/tools/timeline.sh 220 --nu-scans 0 > $O/${T}_timeline.txt 2>&1
head -1 $O/${T}_timeline.txt; python - $O/${T}_timeline.txt <<'PY'
import sys, collections
acc=collections.defaultdict(list)
for ln in open(sys.argv[1]).read().splitlines()[1:]:
    p=ln.split(None,3)
    if len(p)==4:
        try: acc[p[3].strip().split('(')[0][:40]].append(float(p[1]))
        except ValueError: pass
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])): print(f"{k:42s} n={len(v):3d} mean {sum(v)/len(v):6.1f} us  max {max(v):6.1f}")
PY
