#!/bin/bash
# round 6: wave-cooperative leaf matching -- parity, then configs[3] and the headline with it and with the lane-by-lane walk (IMMESH_MATCH_SEQ) on one box
R=$GRAFT_REPO_ROOT
cd $R; timeout 1500 python -m pytest tests/test_gpu_registration.py tests/test_gpu_residency.py tests/test_gpu_parity_fullsize.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | head -12
cd /tmp && export TMPDIR=/tmp
c4() { timeout 300 python $R/bench.py --gpus 1 --config velodyne --map-scans 50 --steps 20 --warmup 3 --cpu-seconds 0 --profile-scans 8 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step']); print('   ', {k.split('(')[0][:28]: v for k, v in d.get('kernels_ms_per_scan', {}).items() if v > 0.05})"; }
one() { timeout 200 python $R/bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARM:-5} --nu-scans 0 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do c4 c4_coop; IMMESH_MATCH_SEQ=1 c4 c4_seq; done
for rep in 1 2; do IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_head.so one head20; one coop20; IMMESH_MATCH_SEQ=1 one seq20; done
export STEPS=500 WARM=20
for rep in 1 2; do IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_head.so one head500; one coop500; IMMESH_MATCH_SEQ=1 one seq500; done
