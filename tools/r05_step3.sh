#!/bin/bash
# round 5, step 3: C4-map parity test (shadow map), write-back attribution calibration, configs[3] as C4 with its CPU leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_parity_fullsize.py -q -k "first_50" -s > $O/c4_test.log 2>&1; grep -E "parity\]|passed|failed|^E" $O/c4_test.log | head
bash tools/fetch_calib.sh > /dev/null 2>&1; cp $R/gpurun_out/fetch_calib.txt $O/write_calib.txt; grep -E "small_dirty|noop_after|stream_write" $O/write_calib.txt
cd /tmp && export TMPDIR=/tmp
timeout 500 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --profile-scans 0 --extra-configs 0 --config velodyne --cpu-seconds 8 2>$O/leg_kitti.err | grep '^{' | tail -1 > $O/leg_kitti.json
python - $O/leg_kitti.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], d["config"]["map_root_voxels"], d["config"]["registration_map"], json.dumps(d.get("cpu_baseline"))[:600])
PY
grep "C4 map" $O/leg_kitti.err
