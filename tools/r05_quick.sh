#!/bin/bash
# round 5: a targeted slice of the GPU tier (argument: pytest selection), log under gpurun_out/r05/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout ${T:-900} python -m pytest "$@" -q > $O/quick.log 2>&1; echo "pytest rc $?" >> $O/quick.log
tail -25 $O/quick.log
