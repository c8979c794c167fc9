#!/bin/bash
# A/B on ONE box: bench.py under different environment switches (and, if present, against a library built from another commit).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
one() { timeout 80 python $R/bench.py --cpu-seconds 0 --extra-configs 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['stages_ms_serial'], d['scan_thread_ms']); print('   ', {k.split('(')[0][:24]: v for k, v in d['kernels_ms_per_scan'].items() if v > 0.012})"; }
one base
IMMESH_PARTITION=160 one part160
IMMESH_PARTITION=192 one part192
IMMESH_PARTITION=128 one part128
if [ -f $R/immesh_amd/csrc/libimmesh_hip_head.so ]; then
  cp $R/immesh_amd/csrc/libimmesh_hip.so /tmp/cur.so; cp $R/immesh_amd/csrc/libimmesh_hip_head.so $R/immesh_amd/csrc/libimmesh_hip.so
  one alt
  IMMESH_MESH_GRID_DIV=2 one alt+griddiv2
  cp /tmp/cur.so $R/immesh_amd/csrc/libimmesh_hip.so
fi
