#!/bin/bash
# the two new bench legs on one box: dense mesh seed (SURVEY C3 density) and the 500-scan steady state.  usage: tools/r03_legs.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-legs}; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
if [ -n "$TESTS" ]; then timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -8; fi
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --profile-scans 5 --dense-mesh 1 --gpu-scans 1 2>$O/${T}_dense.err | grep '^{' | tail -1 > $O/${T}_dense.json
tail -4 $O/${T}_dense.err
python -c "
import json; d=json.load(open('$O/${T}_dense.json')); print('DENSE', d['value'], d['ms_per_step'], d['stages_ms_serial']); print(d['mesh_seed']); print(d['n_u']); print(d['counters_per_scan']); print(d['kernels_ms_per_scan'])"
timeout 400 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --profile-scans 0 --gpu-scans 1 --steps 500 --warmup 20 --nu-scans 5 2>$O/${T}_steady.err | grep '^{' | tail -1 > $O/${T}_steady.json
tail -2 $O/${T}_steady.err
python -c "
import json; d=json.load(open('$O/${T}_steady.json')); print('STEADY', d['value'], d['ms_per_step'], d['scan_thread_ms'], d['pose_err_m']); print(d['n_u']); print(d['counters_per_scan'])"
