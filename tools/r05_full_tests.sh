#!/bin/bash
# round 5: the whole GPU tier, every failure reported (no -x)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/gputests_full.log 2>&1; echo "pytest rc $?" >> $O/gputests_full.log
tail -40 $O/gputests_full.log
