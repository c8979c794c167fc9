#!/bin/bash
# the default bench line exactly as the driver runs it (no tests); usage: tools/r03_drv.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-drv}; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( time timeout 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>$O/$T.err | grep '^{' | tail -1 > $O/$T.json ) 2>&1 | tail -4
python -c "
import json; d=json.load(open('$O/$T.json'))
print('BENCH', d['value'], d['ms_per_step'], d['stages_ms_serial'], d['scan_thread_ms'], d['config']['mesh_map'])
print('roofline', {k: v for k, v in d['roofline'].items() if k not in ('source', 'traffic_source', 'counters_of_the_profiled_scans')})
print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k != 'sample'})
for k, v in d['extra'].items(): print(' extra', k[:64], {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'scan_thread_ms', 'error', 'steps')}, (v.get('cpu_baseline') or {}).get('value'))
print(d.get('n_u')); print(d['counters_per_scan']); print(d['mesh_seed'])"
tail -4 $O/$T.err
