#!/bin/bash
# round 6: configs[3] (velodyne.yaml, scans of varying size) under variants -- the phase graphs keyed by the rounded candidate count
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python $R/bench.py --gpus 1 --config velodyne --steps 20 --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --nu-scans 0 2>/tmp/err.txt | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; grep '^\[mesh marks\]' /tmp/err.txt | cut -c1-420; }
export IMMESH_DEBUG_WAITS=1
for rep in 1 2; do
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; one "$v" )
done
done
