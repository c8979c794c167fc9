#!/bin/bash
# VoxelGrid in the loop: graph / direct launches / deferred launch, with the scan thread's time in begin and end
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_downsample.py tests/test_gpu_registration.py -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^\s+File|pluggy|_pytest|Extension modules" | tail -5
cd /tmp && export TMPDIR=/tmp
run() {
  env $1 timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --profile-scans ${3:-0} --nu-scans 0 --steps 200 $2 2>/tmp/q.err | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [$1 $2]', d['value'], d['ms_per_step'], d['scan_thread_ms'], {k:v for k,v in (d.get('kernels_ms_per_scan') or {}).items() if k.startswith('ds_')})"
  grep 'VoxelGrid on the scan thread' /tmp/q.err
}
run A=1 ""
run A=1 "--device-downsample 1" 3
run A=1 "--mesh 0 --device-downsample 1"
run A=1 ""
run A=1 "--device-downsample 1"
