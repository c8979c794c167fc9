#!/bin/bash
# round 5, first GPU call: the GPU tier on the round's correctness changes, then the default bench line and a steady-state timeline as the round's baseline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests_1.log 2>&1; echo "pytest rc $?" >> $O/gputests_1.log
tail -5 $O/gputests_1.log
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_base.err | grep '^{' | tail -1 > $O/bench_base.json
cut -c1-400 $O/bench_base.json
bash $R/tools/timeline.sh 220 --nu-scans 0 > $O/timeline_base.txt 2>&1
tail -40 $O/timeline_base.txt
