#!/bin/bash
# the driver's round-end sequence on one box: GPU test tier, smoke, the default bench line (as the driver runs it).  usage: tools/r03_full.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-full}; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) 2>&1 | tail -25
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
( time timeout 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>$O/$T.err | grep '^{' | tail -1 > $O/$T.json ) 2>&1 | tail -4
python -c "
import json; d=json.load(open('$O/$T.json'))
print('BENCH', d['value'], d['ms_per_step'], d['stages_ms_serial'], d['scan_thread_ms'])
print('roofline', d['roofline'])
print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k != 'sample'})
for k, v in d['extra'].items(): print(' extra', k[:60], {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'scan_thread_ms', 'error', 'steps')}, (v.get('cpu_baseline') or {}).get('value'))
print(d.get('n_u'))"
tail -5 $O/$T.err
