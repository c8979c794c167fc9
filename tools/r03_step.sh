#!/bin/bash
# one development step on the GPU box: GPU test tier, default bench line, IMMESH_DEBUG phase tables.  usage: tools/r03_step.sh <tag> [pytest args]
R=$GRAFT_REPO_ROOT; T=${1:-step}; shift; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 2>$O/$T.err | grep '^{' | tail -1 > $O/$T.json
python -c "
import json; d=json.load(open('$O/$T.json')); print('BENCH', d['value'], d['ms_per_step'], d['stages_ms_serial'], d['scan_thread_ms']); print(d['kernels_ms_per_scan']); print(d['counters_per_scan'])"
tail -3 $O/$T.err
IMMESH_DEBUG=1 timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 3 --profile-scans 0 --async-mesh 0 2>$O/${T}_dbg.err | grep '^{' | tail -1 | cut -c1-200
grep -E '^\[re' $O/${T}_dbg.err | tail -6
