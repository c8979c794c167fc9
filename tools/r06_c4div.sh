#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
c4() { timeout 300 python $R/bench.py --gpus 1 --config velodyne --map-scans 50 --steps 20 --warmup 3 --cpu-seconds 0 --profile-scans 8 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step']); print('   ', {k.split('(')[0][:28]: v for k, v in d.get('kernels_ms_per_scan', {}).items() if v > 0.02})"; }
for rep in 1 2; do
c4 div128; IMMESH_LIST_DIV=32 c4 div32; IMMESH_LIST_DIV=16 c4 div16; IMMESH_LIST_DIV=8 c4 div8
done
