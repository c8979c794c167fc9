#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 200 python $R/bench.py --gpus 1 --steps ${STEPS:-500} --warmup 20 --nu-scans 0 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_head.so one head
  one new
done
