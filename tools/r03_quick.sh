#!/bin/bash
# quick bench-only check (no profile legs); usage: tools/r03_quick.sh [bench args]
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --profile-scans 0 --nu-scans 0 "$@" 2>/tmp/q.err | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['scan_thread_ms'])"
grep "drain\|pre-build" /tmp/q.err
