#!/bin/bash
# round-3 baseline on one box: default bench line, IMMESH_DEBUG phase timers (host-EKF mode), steady-state kernel timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 2>$O/base.err | grep '^{' | tail -1 > $O/base.json
python -c "
import json; d=json.load(open('$O/base.json')); print('BASE', d['value'], d['ms_per_step'], d['stages_ms_serial'], d['scan_thread_ms']); print(d['kernels_ms_per_scan']); print(d['counters_per_scan'])"
grep 'pre-build' $O/base.err
IMMESH_DEBUG=1 timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 3 --profile-scans 0 --async-mesh 0 2>$O/dbg.err | grep '^{' | tail -1 > $O/dbg.json
grep -E '^\[re' $O/dbg.err | tail -12
bash $R/tools/timeline.sh 260 > $O/timeline.txt 2>&1; head -3 $O/timeline.txt; tail -130 $O/timeline.txt
