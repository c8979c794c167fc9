#!/bin/bash
# round 6: one step = the whole GPU tier (every failure reported) + the quick perf check; usage: tools/r06_step.sh <tag> [pytest -k expression]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; T=${1:-step}
cd $R
if [ -n "$2" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$2" > $O/${T}_tests.log 2>&1; else timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/${T}_tests.log 2>&1; fi
echo "pytest rc $?" >> $O/${T}_tests.log
tail -25 $O/${T}_tests.log
bash $R/tools/r06_perf.sh $T
