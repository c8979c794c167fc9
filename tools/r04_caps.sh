#!/bin/bash
# experiment: capacity of the mesh map / per-scan scratch vs kernel times (TLB reach of the hash tables)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for a in "--mesh-cap-log2 24" "--mesh-cap-log2 21" "--mesh-cap-log2 21 --cap-scan-points 400000"; do
timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --profile-scans 3 --nu-scans 0 --steps 40 $a 2>/tmp/q.err | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BENCH [$a]', d['value'], d['ms_per_step'], d['scan_thread_ms'], d['stages_ms_serial']); print({k:v for k,v in d['kernels_ms_per_scan'].items() if v>0.009})"
done
