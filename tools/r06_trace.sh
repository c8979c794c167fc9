#!/bin/bash
# round 6: the in-kernel traces (IMMESH_DEBUG) of the asynchronous run under environment variants; usage: tools/r06_trace.sh <tag> "A=1 B=2" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; T=$1; shift
cd /tmp && export TMPDIR=/tmp
: > $O/${T}_traces.txt
for v in "$@"; do
  ( [ "$v" != "-" ] && export $v
    IMMESH_DEBUG=1 IMMESH_TRACE_FILE=/tmp/trace_x.bin timeout 200 python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 3 --profile-scans 0 --nu-scans 0 --async-mesh 1 2>/tmp/dbg_x.err > /dev/null
    { echo "== $v"; grep -E '^\[(re|slow)' /tmp/dbg_x.err | tail -3; python $R/tools/trace_report.py /tmp/trace_x.bin blocks; } >> $O/${T}_traces.txt )
done
grep -E "^==|kernel entry|per block|pass 0|epilogue|replay_fused:|\[replay_list" $O/${T}_traces.txt
