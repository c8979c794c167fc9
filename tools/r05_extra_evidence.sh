#!/bin/bash
# round 5: two more measurements on the final sources -- (1) the configs[4] dry run with 30 timed scans per rank (the default leg times 10: a short sample
# for a median), (2) rocprofv3 kernel stats of configs[3] as SURVEY 8(d) C4 (map from the first 50 scans)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --gpus 1 --dry-run-rank -2 --pts 500000 --map-voxels 50e6 --steps 30 --warmup 3 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2> $O/dry30.err | grep '^{' | tail -1 > $O/dry_run_all_ranks_30.json
cut -c1-300 $O/dry_run_all_ranks_30.json
rm -rf /tmp/rp_c4; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_c4 -- python $R/bench.py --config velodyne --steps 20 --warmup 5 --cpu-seconds 0 --extra-configs 0 > /tmp/rp_c4.log 2>&1
cp $(find /tmp/rp_c4 -name '*kernel_stats.csv' | head -1) $O/c4_kernel_stats.csv
grep '^{' /tmp/rp_c4.log | tail -1 > $O/c4_bench_under_rocprof.json
head -25 $O/c4_kernel_stats.csv | cut -c1-200
