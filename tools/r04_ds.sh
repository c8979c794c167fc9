#!/bin/bash
# VoxelGrid step: the down-sampling tests, then the raw-scan-to-pose leg (device VoxelGrid inside the loop) beside the default headline
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_downsample.py tests/test_decode.py tests/test_undistort.py -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^\s+File|pluggy|_pytest|Extension modules" | tail -15
cd /tmp && export TMPDIR=/tmp
for a in "--device-downsample 1" ""; do
timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --profile-scans 3 --nu-scans 0 $a 2>/tmp/q.err | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [$a]', d['value'], d['ms_per_step'], d['scan_thread_ms']); print({k:v for k,v in d['kernels_ms_per_scan'].items() if k.startswith('ds_') or 'rocprim' in k or 'sort' in k.lower()})"
done
