#!/bin/bash
# round 6: how often does the driver's 20-scan run come out slow, and where is the time?  N repeats per variant, value + drain line + slowest calls
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --nu-scans 0 2>/tmp/err.txt | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'), end=' | ')"; grep 'drain after' /tmp/err.txt | sed 's/\[bench\] drain after the last scan: //'; }
N=${N:-12}
for rep in $(seq 1 $N); do
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; one "$v" )
done
done
