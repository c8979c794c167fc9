#!/bin/bash
# round 6, first GPU call: is hipExtAnyOrderLaunch honoured on gfx950 + the baseline perf of the round-5 sources on today's box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
$R/tools/anyorder_test.bin > $O/anyorder.txt 2>&1; cat $O/anyorder.txt
bash $R/tools/r06_perf.sh base
