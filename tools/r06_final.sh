#!/bin/bash
# round 6: the whole GPU tier, then everything under profiles/ regenerated on the same sources (tools/refresh_profiles.sh r06), the mesher's phase marks
# in the unprofiled pipeline (IMMESH_DEBUG_WAITS) for the shipped arrangement and for round 5's (IMMESH_NO_SPLIT=1 IMMESH_MESH_ROOM=2), configs[3] under rocprofv3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O $R/gpurun_out/profiles_new
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests_final.log 2>&1; echo "pytest rc $?" >> $O/gputests_final.log
tail -4 $O/gputests_final.log
bash $R/tools/refresh_profiles.sh r06 > $O/refresh.log 2>&1
tail -30 $O/refresh.log | cut -c1-400
bash $R/tools/r06_marks.sh - IMMESH_SPLIT=1 "IMMESH_NO_SPLIT=1 IMMESH_MESH_ROOM=2" > $R/gpurun_out/profiles_new/r06_marks_pipeline.txt 2>&1
N=6 bash $R/tools/r06_outliers.sh - IMMESH_SPLIT=1 "IMMESH_NO_SPLIT=1 IMMESH_MESH_ROOM=2" > $R/gpurun_out/profiles_new/r06_driver_run_repeats.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_c4; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_c4 -- python $R/bench.py --config velodyne --steps 20 --warmup 5 --cpu-seconds 0 --extra-configs 0 > /tmp/rp_c4.log 2>&1
cp $(find /tmp/rp_c4 -name '*kernel_stats.csv' | head -1) $R/gpurun_out/profiles_new/r06_c4_kernel_stats.csv
grep '^{' /tmp/rp_c4.log | tail -1 > $R/gpurun_out/profiles_new/r06_c4_bench_under_rocprof.json
cp $O/gputests_final.log $R/gpurun_out/profiles_new/r06_gpu_tier.log
ls -la $R/gpurun_out/profiles_new
