#!/bin/bash
# round 6: A/B of the list kernel beside / behind the fused kernel on ONE box + the in-kernel traces of the asynchronous run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; T=${1:-diag}
cd /tmp && export TMPDIR=/tmp
one() { timeout 200 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --profile-scans 0 --extra-configs 0 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('scan_thread_ms'))"; }
one beside
IMMESH_LIST_SERIAL=1 one serial
one beside
IMMESH_LIST_SERIAL=1 one serial
for mode in 1; do
  IMMESH_DEBUG=1 IMMESH_TRACE_FILE=/tmp/trace_$mode.bin timeout 200 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 3 --profile-scans 0 --nu-scans 0 --async-mesh $mode 2>/tmp/dbg_$mode.err > /dev/null
  { echo "== IMMESH_DEBUG phase timers, --async-mesh $mode"; grep -E '^\[(re|del|slow|knn)' /tmp/dbg_$mode.err | tail -6;
    echo "== per-wavefront traces of the last launches, --async-mesh $mode"; python $R/tools/trace_report.py /tmp/trace_$mode.bin; } > $O/${T}_phase_tables.txt
  IMMESH_LIST_SERIAL=1 IMMESH_DEBUG=1 IMMESH_TRACE_FILE=/tmp/trace_s$mode.bin timeout 200 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 3 --profile-scans 0 --nu-scans 0 --async-mesh $mode 2>/tmp/dbg_s$mode.err > /dev/null
  { echo "== SERIAL LIST: IMMESH_DEBUG phase timers, --async-mesh $mode"; grep -E '^\[(re|del|slow|knn)' /tmp/dbg_s$mode.err | tail -6;
    echo "== per-wavefront traces of the last launches, --async-mesh $mode"; python $R/tools/trace_report.py /tmp/trace_s$mode.bin; } >> $O/${T}_phase_tables.txt
done
cat $O/${T}_phase_tables.txt
