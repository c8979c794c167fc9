#!/bin/bash
# round 6: the admission on fresh ground -- mesh_append_resolve_kernel over the bench's seeding packages (rocprofv3 stats) and the scan-0-seeded leg, new library vs HEAD~'s
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_mesher.py tests/test_gpu_sharded.py -m gpu -q -x -k "not 8-3" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for lib in new head; do
  [ $lib = head ] && export IMMESH_HIP_LIBRARY=$R/immesh_amd/csrc/libimmesh_hip_head.so || unset IMMESH_HIP_LIBRARY
  for rep in 1 2; do
    timeout 200 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 --dense-mesh 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib scan-0-seeded mesh map:', d['value'], d['ms_per_step'])"
  done
  rm -rf /tmp/adm_$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/adm_$lib -- python $R/bench.py --cpu-seconds 0 --steps 20 --warmup 5 --profile-scans 0 --extra-configs 0 > /dev/null 2>&1
  f=$(find /tmp/adm_$lib -name '*kernel_stats.csv' | head -1)
  echo "== $lib: kernel stats of the default run (seeding packages included)"; grep -E "Name|mesh_append_resolve|mesh_append_prepare|mesh_append_finish" $f | cut -c1-200
  cp $f $O/admission_${lib}_kernel_stats.csv
done
