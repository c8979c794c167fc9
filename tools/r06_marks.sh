#!/bin/bash
# round 6 (second session): phase marks of the mesher's jobs in the UNPROFILED steady state (IMMESH_DEBUG_WAITS: kernel-entry times relative to the job's
# start, publish-to-publish period, the scan thread's wait for a world buffer); variants as arguments ("-" = none), each run twice
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python $R/bench.py --gpus 1 --steps $2 --warmup $3 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --nu-scans 0 $4 2>/tmp/err.txt | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'steps $2', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; grep '^\[mesh marks\]' /tmp/err.txt; }
export IMMESH_DEBUG_WAITS=1
for rep in 1 2; do
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; one "$v" 500 20 "--gpu-scans 1"; [ $rep = 1 ] && one "$v" 20 5 )
done
done
