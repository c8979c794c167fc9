#!/bin/bash
# round 6: A/B of the product library against variant builds (immesh_amd/csrc/libimmesh_hip_<tag>.so), 20-scan run x4 + 500-scan run; usage: tools/r06_ab_lib.sh <tag> [<tag> ...]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python $R/bench.py --gpus 1 $2 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --nu-scans 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; }
for rep in 1 2 3 4; do
  one product "--steps 20 --warmup 5"
  for t in "$@"; do IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_$t.so one $t "--steps 20 --warmup 5"; done
done
one product500 "--gpu-scans 1 --steps 500 --warmup 20"
for t in "$@"; do IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_$t.so one ${t}500 "--gpu-scans 1 --steps 500 --warmup 20"; done
