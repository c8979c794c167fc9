#!/bin/bash
# Regenerates everything under profiles/ on the GPU box (one MI355X).  Outputs go to gpurun_out/profiles_new/ (merged back by gpurun);
# copy them into profiles/ afterwards.  Every leg runs under its own timeout.  usage: tools/refresh_profiles.sh [round tag, default r02]
R=$GRAFT_REPO_ROOT; T=${1:-r02}; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { grep '^{' | tail -1; }
timeout 400 python $R/bench.py 2>/dev/null | line > $O/${T}_bench_full.json
# rocprofv3 kernel stats of the default command (CPU leg and child runs skipped: they launch nothing of interest)
rm -rf /tmp/rp_stats; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- python $R/bench.py --cpu-seconds 0 --extra-configs 0 > /tmp/rp_stats.log 2>&1
cp $(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1) $O/${T}_full_kernel_stats.csv
grep '^{' /tmp/rp_stats.log | tail -1 > $O/${T}_bench_under_rocprof.json
# HBM traffic: one counter per pass, the same command in serial mode (per-launch byte counts do not depend on the overlap)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$ctr; timeout 180 rocprofv3 --pmc $ctr --output-format csv -d /tmp/rp_$ctr -- python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 2 --profile-scans 0 --async-mesh 0 > /tmp/rp_$ctr.log 2>&1
done
python $R/tools/pmc_traffic.py $(find /tmp/rp_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/rp_WRITE_SIZE -name '*counter_collection.csv' | head -1) $O/traffic_${T}.json
ls -la $O; for f in $O/${T}_bench_*.json; do echo $f; cut -c1-200 $f; done
