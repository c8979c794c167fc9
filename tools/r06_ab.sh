#!/bin/bash
# round 6: A/B on ONE box: the working tree's library (list kernel beside / behind the fused kernel) against the library built from HEAD~ (libimmesh_hip_head.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { timeout 200 python $R/bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; }
for rep in 1 2; do
  IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_head.so one head
  one new
  for v in "$@"; do export $v; one "$v"; unset ${v%%=*}; done
done
