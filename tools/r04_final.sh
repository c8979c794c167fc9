#!/bin/bash
# round-end refresh, the short form: the default bench line (with roofline.traffic from the committed file).   usage: tools/r04_final.sh [stats]
# with "stats": also rocprofv3 kernel stats of the same command and the steady-state timelines with and without the VoxelGrid in the loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>$O/r04_bench_full.err | grep '^{' | tail -1 > $O/r04_bench_full.json
python -c "
import json; d=json.load(open('$O/r04_bench_full.json'))
print('BENCH', d['value'], d['ms_per_step'], d['scan_thread_ms']); print('roofline', {k: v for k, v in d['roofline'].items() if k in ('achieved', 'frac', 'traffic', 'avg_launch_ms')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['reference_threading']['stages_ms_p50'], (d['cpu_baseline'].get('all_cores') or {}).get('scans_per_s'))
for k, v in d['extra'].items(): print(' extra', k[:60], v.get('value'), v.get('ms_per_step'), v.get('error'), (v.get('cpu_baseline') or {}).get('value'))"
tail -3 $O/r04_bench_full.err
[ "$1" = stats ] || exit 0
rm -rf /tmp/rp_stats; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- python $R/bench.py --cpu-seconds 0 --extra-configs 0 > /tmp/rp_stats.log 2>&1
cp $(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1) $O/r04_full_kernel_stats.csv
grep '^{' /tmp/rp_stats.log | tail -1 > $O/r04_bench_under_rocprof.json
bash $R/tools/timeline.sh 220 --nu-scans 0 > $O/r04_timeline_steady_state.txt 2>&1
bash $R/tools/timeline.sh 120 --nu-scans 0 --device-downsample 1 > $O/r04_timeline_voxelgrid_in_loop.txt 2>&1
