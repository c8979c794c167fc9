#!/bin/bash
# Regenerates everything under profiles/ on the GPU box (one MI355X).  Outputs go to gpurun_out/profiles_new/ (merged back by gpurun);
# copy them into profiles/ afterwards.  Every leg runs under its own timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { grep '^{' | tail -1; }
timeout 300 python $R/bench.py 2>/dev/null | line > $O/r01_bench_full.json
timeout 200 python $R/bench.py --mesh 0 2>/dev/null | line > $O/r01_bench_reg_only.json
timeout 300 python $R/bench.py --config velodyne --steps 20 --warmup 3 2>/dev/null | line > $O/r01_bench_kitti.json
timeout 150 python $R/bench.py --pts 500000 --steps 10 --warmup 2 --profile-scans 2 --cpu-seconds 0 2>/dev/null | line > $O/r01_bench_500k_pts.json
timeout 150 python $R/bench.py --device-downsample 1 --cpu-seconds 0 2>/dev/null | line > $O/r01_bench_device_downsample.json
# rocprofv3 kernel stats of the default command (CPU leg skipped: it launches nothing)
rm -rf /tmp/rp_stats; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- python $R/bench.py --cpu-seconds 0 > /tmp/rp_stats.log 2>&1
cp $(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1) $O/r01_full_kernel_stats.csv
# HBM traffic: one counter per pass
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$ctr; timeout 180 rocprofv3 --pmc $ctr --output-format csv -d /tmp/rp_$ctr -- python $R/bench.py --cpu-seconds 0 --steps 12 --warmup 2 --profile-scans 0 --async-mesh 0 > /tmp/rp_$ctr.log 2>&1
done
python $R/tools/pmc_traffic.py $(find /tmp/rp_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/rp_WRITE_SIZE -name '*counter_collection.csv' | head -1) $O/traffic_r01.json
ls -la $O; for f in $O/r01_bench_*.json; do echo $f; cut -c1-200 $f; done
