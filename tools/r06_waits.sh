#!/bin/bash
# round 6 (second session): who waits for whom?  IMMESH_DEBUG_WAITS prints how long the scan thread stood at mesh_next_world_buffer (= the mesher is the
# bottleneck) and the device time of a mesher job; variants as arguments ("-" = none)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python $R/bench.py --gpus 1 --steps $2 --warmup $3 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --nu-scans 0 $4 2>/tmp/err.txt | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'steps $2', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; grep '^\[mesh\]' /tmp/err.txt; }
export IMMESH_DEBUG_WAITS=1
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; one "$v" 20 5; one "$v" 500 20 "--gpu-scans 1" )
done
