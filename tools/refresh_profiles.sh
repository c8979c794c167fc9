#!/bin/bash
# Regenerates everything under profiles/ on the GPU box (one MI355X).  Outputs go to gpurun_out/profiles_new/ (merged back by gpurun);
# copy them into profiles/ afterwards.  Every leg runs under its own timeout.  usage: tools/refresh_profiles.sh [round tag, default r03]   (SKIP_FULL=1: leg (1) is taken from an earlier run)
R=$GRAFT_REPO_ROOT; T=${1:-r03}; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { grep '^{' | tail -1; }
# (1) the default command, as the driver runs it
if [ -z "$SKIP_FULL" ]; then timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>$O/${T}_bench_full.err | line > $O/${T}_bench_full.json; fi
# (2) rocprofv3 kernel stats of the default command (CPU leg and child runs skipped: they launch nothing of interest)
rm -rf /tmp/rp_stats; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- python $R/bench.py --cpu-seconds 0 --extra-configs 0 > /tmp/rp_stats.log 2>&1
cp $(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1) $O/${T}_full_kernel_stats.csv
grep '^{' /tmp/rp_stats.log | tail -1 > $O/${T}_bench_under_rocprof.json
# (3) HBM traffic: one counter per pass, the same command in serial mode (per-launch byte counts do not depend on the overlap)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$ctr; IMMESH_SERIAL_SAFE=1 timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/rp_$ctr -- python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 2 --profile-scans 0 --nu-scans 0 --async-mesh 0 > /tmp/rp_$ctr.log 2>&1
done
python $R/tools/pmc_traffic.py $(find /tmp/rp_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/rp_WRITE_SIZE -name '*counter_collection.csv' | head -1) $O/traffic_${T}.json
# (4) steady-state kernel timeline (start / duration / queue of every kernel)
bash $R/tools/timeline.sh 220 --nu-scans 0 > $O/${T}_timeline_steady_state.txt 2>&1
# (5) in-kernel phase timers + per-wavefront traces (IMMESH_DEBUG): replay_fused / residual_persistent / delaunay64 phase tables, serial and pipelined
for mode in 0 1; do
  IMMESH_DEBUG=1 IMMESH_TRACE_FILE=/tmp/trace_$mode.bin timeout 200 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 3 --profile-scans 0 --nu-scans 0 --async-mesh $mode 2>/tmp/dbg_$mode.err > /dev/null
  { echo "== IMMESH_DEBUG phase timers, --async-mesh $mode (cycles of the shader clock; timers on: the totals are inflated, the split is what counts)"; grep -E '^\[(re|del|slow|knn)' /tmp/dbg_$mode.err | tail -6;
    echo "== per-wavefront traces of the last launches, --async-mesh $mode"; python $R/tools/trace_report.py /tmp/trace_$mode.bin; } >> $O/${T}_phase_tables.txt
done
# (6) SQ counters (instruction mix, active / waiting) per kernel, serial mode
bash $R/tools/pmc_insts.sh > $O/${T}_pmc_sq_summary.txt 2>&1
ls -la $O; for f in $O/${T}_bench_*.json; do echo $f; cut -c1-300 $f; done
