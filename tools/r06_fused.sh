#!/bin/bash
# round 6 (second session): does a smaller resident grid of replay_fused_kernel (IMMESH_FUSED_WGS) leave the mesher's short phase-B launches room?
# 20-scan headline twice + 500-scan steady state per variant, the pure pose chain (--mesh 0) for reference
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python $R/bench.py --gpus 1 --steps $2 --warmup $3 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --nu-scans 0 $4 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'steps $2', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; }
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; one "$v" 20 5; one "$v" 20 5; one "$v" 500 20 "--gpu-scans 1" )
done
one mesh0 500 20 "--gpu-scans 1 --mesh 0"
