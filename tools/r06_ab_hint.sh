#!/bin/bash
# round 6: A/B of the epilogue's slot hint (libimmesh_hip_nohint.so = the same sources with -DIMMESH_EPI_HINT=0) + the mesher's grid divisor, 20-scan run
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 300 python $R/bench.py --gpus 1 $(if [ -z "$2" ]; then echo "--steps 20 --warmup 5"; fi) --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --nu-scans 0 $2 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; }
for rep in 1 2 3 4; do
  one hint
  IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_nohint.so one nohint
  IMMESH_MESH_GRID_DIV=4 one griddiv4
done
one hint500 "--gpu-scans 1 --steps 500 --warmup 20"; IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_nohint.so one nohint500 "--gpu-scans 1 --steps 500 --warmup 20"
