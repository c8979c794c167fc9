#!/bin/bash
# round 5, last confirmation on the final sources: the whole GPU tier, then the default command as the driver runs it (the bench line with roofline.traffic
# read from the committed traffic_r05.json, whose fingerprint must match this build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests_confirm.log 2>&1; echo "pytest rc $?" >> $O/gputests_confirm.log
tail -4 $O/gputests_confirm.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_confirm.err | grep '^{' | tail -1 > $O/bench_confirm.json
cut -c1-400 $O/bench_confirm.json; tail -3 $O/bench_confirm.err
