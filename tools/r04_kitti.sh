#!/bin/bash
# configs[3] (KITTI-shaped, velodyne.yaml): headline + per-kernel times + the replay_list debug counters
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --config velodyne --cpu-seconds 0 --extra-configs 0 --profile-scans 4 --nu-scans 0 --steps 20 "$@" 2>/tmp/q.err | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['scan_thread_ms'], d['stages_ms_serial']); print(d['kernels_ms_per_scan']); print(d['counters_per_scan']); print(d['config']['n_ds_mean'], d['config']['map_root_voxels'])"
tail -2 /tmp/q.err
IMMESH_DEBUG=1 timeout 200 python $R/bench.py --config velodyne --cpu-seconds 0 --extra-configs 0 --steps 8 --warmup 3 --profile-scans 0 --nu-scans 0 --async-mesh 0 2>/tmp/dbg.err > /dev/null
grep -E '^\[(re|slow)' /tmp/dbg.err | tail -4
