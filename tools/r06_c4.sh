#!/bin/bash
# round 6: configs[3] (C4: velodyne.yaml, map from the stream's first 50 scans) with and without the moment records + parity of the deep-octree cases + the headline A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R; timeout 1200 python -m pytest tests/test_gpu_registration.py tests/test_gpu_parity_fullsize.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -12
cd /tmp && export TMPDIR=/tmp
c4() { timeout 300 python $R/bench.py --gpus 1 --config velodyne --map-scans 50 --steps 20 --warmup 3 --cpu-seconds 0 --profile-scans 8 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step']); print('   ', {k.split('(')[0][:28]: v for k, v in d.get('kernels_ms_per_scan', {}).items() if v > 0.01})"; }
c4 moments; IMMESH_NO_MOMENTS=1 c4 no-moments; c4 moments; IMMESH_NO_MOMENTS=1 c4 no-moments
bash $R/tools/r06_ab.sh
