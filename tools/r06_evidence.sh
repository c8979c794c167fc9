#!/bin/bash
# round 6: two more pieces of evidence on the final sources -- (1) the kernel timeline of the 500-scan leg (the worker's deep arrangement: phase A, the
# triangulations on the third stream and phase B of three different jobs side by side), (2) the pose chain alone (--mesh 0) over the driver's 20 scans and over 500
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_new; mkdir -p $O
bash $R/tools/timeline.sh 200 --nu-scans 0 --gpu-scans 1 --steps 500 --warmup 20 > $O/r06_timeline_500_scans_deep.txt 2>&1
head -3 $O/r06_timeline_500_scans_deep.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --mesh 0 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 > $O/r06_bench_mesh0_20.json
timeout 300 python $R/bench.py --gpus 1 --gpu-scans 1 --steps 500 --warmup 20 --mesh 0 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 > $O/r06_bench_mesh0_500.json
cut -c1-200 $O/r06_bench_mesh0_20.json; cut -c1-200 $O/r06_bench_mesh0_500.json
