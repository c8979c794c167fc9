#!/bin/bash
# round 5, step 2: the C4-map parity test again, the WRITE_SIZE calibration, configs[3] as C4 and the drop-in legs (reference mirror / stand-in)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_parity_fullsize.py -q -k "first_50" > $O/c4_test.log 2>&1; tail -3 $O/c4_test.log
bash tools/fetch_calib.sh > /dev/null 2>&1; cp $R/gpurun_out/fetch_calib.txt $O/write_calib.txt; cat $O/write_calib.txt
cd /tmp && export TMPDIR=/tmp
for leg in "kitti --config velodyne --steps 20 --warmup 5 --cpu-seconds 8" "kitti_scan0 --config velodyne --map-scans 1 --steps 20 --warmup 5" "dropin_ref --dropin-shim 1 --dropin-mirror ref" "dropin_stub --dropin-shim 1 --dropin-mirror stub"; do
  set -- $leg; name=$1; shift
  timeout 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 "$@" 2>$O/leg_$name.err | grep '^{' | tail -1 > $O/leg_$name.json
  python - $O/leg_$name.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], d["config"].get("map_root_voxels"), json.dumps(d.get("drop_in_shim"))[:900], json.dumps(d.get("cpu_baseline",{}).get("value")))
PY
done
