#!/bin/bash
# the driver's command three times: headline, host percentiles and the slowest calls of the timed loop
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --extra-configs 0 --profile-scans 0 --nu-scans 0 2>/tmp/q.err | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['scan_thread_ms'])"
grep 'drain after' /tmp/q.err | cut -c1-300
done
