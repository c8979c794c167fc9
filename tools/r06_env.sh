#!/bin/bash
# round 6: bench under environment variants on ONE box; usage: tools/r06_env.sh "A=1 B=2" "C=3" ...   ("-" = no variables)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 200 python $R/bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; }
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then one base; else ( export $v; one "$v" ); fi
  done
done
