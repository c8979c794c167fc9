#!/bin/bash
# round 6: registration / map-update parity subset + A/B against HEAD's library + traces; usage: tools/r06_quick.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; T=${1:-quick}
cd $R
timeout 900 python -m pytest tests/test_gpu_registration.py tests/test_gpu_residency.py tests/test_gpu_parity_fullsize.py -m gpu -q -x > $O/${T}_tests.log 2>&1; echo "pytest rc $?" >> $O/${T}_tests.log
tail -5 $O/${T}_tests.log
bash $R/tools/r06_ab.sh
STEPS=500 bash $R/tools/r06_ab.sh | head -2
bash $R/tools/r06_trace.sh $T -
