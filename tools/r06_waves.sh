#!/bin/bash
# round 6: residual_persistent_kernel with 1 / 2 / 4 wavefronts per workgroup -- parity subset, then scans/s (20 scans + 500-scan steady state) on one box, against HEAD~'s library
R=$GRAFT_REPO_ROOT
cd $R; timeout 1200 python -m pytest tests/test_gpu_registration.py tests/test_gpu_residency.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head
cd /tmp && export TMPDIR=/tmp
one() { timeout 200 python $R/bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARM:-5} --nu-scans 0 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_head.so one head20
  IMMESH_RP_WAVES=4 one w4_20; IMMESH_RP_WAVES=2 one w2_20; IMMESH_RP_WAVES=1 one w1_20
done
export STEPS=500 WARM=20
for rep in 1 2; do
  IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_head.so one head500
  IMMESH_RP_WAVES=4 one w4_500; IMMESH_RP_WAVES=2 one w2_500; IMMESH_RP_WAVES=1 one w1_500
done
