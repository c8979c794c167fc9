#!/bin/bash
# one development step on the GPU box.  usage: tools/r04_step.sh <tag>   env: TESTS="paths / -k expr" (default: whole gpu tier), SKIP_TESTS=1,
# TRAFFIC=1 (two rocprofv3 --pmc passes -> gpurun_out/r04/traffic_<tag>.json), BENCH_ARGS="..."
R=$GRAFT_REPO_ROOT; T=${1:-step}; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then ( time timeout 1500 python -m pytest ${TESTS:-tests} -m gpu -q --durations=6 -p no:cacheprovider ) > $O/${T}_tests.log 2>&1; grep -E '^(FAILED|ERROR)|passed|failed|^E  ' $O/${T}_tests.log | head -40; fi
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --cpu-seconds 0 --extra-configs 0 $BENCH_ARGS 2>$O/$T.err | grep '^{' | tail -1 > $O/$T.json
python -c "
import json; d=json.load(open('$O/$T.json')); print('BENCH', d['value'], d['ms_per_step'], d['stages_ms_serial'], d['scan_thread_ms']); print(d['kernels_ms_per_scan']); print(d['counters_per_scan'])"
tail -3 $O/$T.err
if [ -n "$TRAFFIC" ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rp_$ctr; IMMESH_SERIAL_SAFE=1 timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/rp_$ctr -- python $R/bench.py --cpu-seconds 0 --extra-configs 0 --steps 12 --warmup 2 --profile-scans 0 --nu-scans 0 --async-mesh 0 $BENCH_ARGS > /tmp/rp_$ctr.log 2>&1
  done
  python $R/tools/pmc_traffic.py $(find /tmp/rp_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/rp_WRITE_SIZE -name '*counter_collection.csv' | head -1) $O/traffic_$T.json
fi
