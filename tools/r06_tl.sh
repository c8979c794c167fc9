#!/bin/bash
# round 6: steady-state kernel timelines (tools/timeline.sh) under environment switches; usage: tools/r06_tl.sh <tag> [ENV=1 ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; T=$1; shift
env "$@" bash $R/tools/timeline.sh 120 --nu-scans 0 > $O/${T}_timeline.txt 2>&1
head -70 $O/${T}_timeline.txt
