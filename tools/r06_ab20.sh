#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 200 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; }
for rep in 1 2 3 4; do
  IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_head.so one r05
  IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_commit.so one commit
  one worktree
done
