#!/bin/bash
# round 6: the driver's 20-scan run under variants, three repeats each, plus the pose chain alone (--mesh 0)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --nu-scans 0 $2 2>/tmp/err.txt | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; grep 'drain after' /tmp/err.txt | sed 's/;.*//'; }
for rep in 1 2 3; do
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v; one "$v" )
done
done
one mesh0 "--mesh 0"; one mesh0 "--mesh 0"
