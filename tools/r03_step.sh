#!/bin/bash
# one development step on the GPU box: GPU test tier, default bench line, IMMESH_DEBUG phase tables + per-wavefront traces.
# usage: tools/r03_step.sh <tag> [pytest args]   (SKIP_TESTS=1 skips the test tier)
R=$GRAFT_REPO_ROOT; T=${1:-step}; shift; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest ${TESTS:-tests} -m gpu -x -q "$@" 2>&1 | tail -15; fi
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 2>$O/$T.err | grep '^{' | tail -1 > $O/$T.json
python -c "
import json; d=json.load(open('$O/$T.json')); print('BENCH', d['value'], d['ms_per_step'], d['stages_ms_serial'], d['scan_thread_ms']); print(d['kernels_ms_per_scan']); print(d['counters_per_scan'])"
tail -3 $O/$T.err
if [ -n "$SMALL_MAP" ]; then
  timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --map-voxels 1e6 2>/dev/null | grep '^{' | tail -1 > $O/${T}_small.json
  python -c "
import json; d=json.load(open('$O/${T}_small.json')); print('SMALL-MAP (1M voxels)', d['value'], d['ms_per_step'], d['stages_ms_serial']); print(d['kernels_ms_per_scan'])"
fi
for mode in 0 1; do
  IMMESH_DEBUG=1 IMMESH_TRACE_FILE=/tmp/trace_$mode.bin timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 3 --profile-scans 0 --async-mesh $mode 2>$O/${T}_dbg$mode.err | grep '^{' | tail -1 | cut -c1-160
  grep -E '^\[re' $O/${T}_dbg$mode.err | tail -2; grep -E '^\[(del|slow|knn)' $O/${T}_dbg$mode.err | tail -4
  echo "--- trace, async-mesh $mode"; python $R/tools/trace_report.py /tmp/trace_$mode.bin
done
