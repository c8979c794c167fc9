#!/bin/bash
# round 5: the whole GPU tier, then everything under profiles/ regenerated on the same sources (tools/refresh_profiles.sh r05)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests_final.log 2>&1; echo "pytest rc $?" >> $O/gputests_final.log
tail -4 $O/gputests_final.log
bash $R/tools/refresh_profiles.sh r05 > $O/refresh.log 2>&1
tail -30 $O/refresh.log | cut -c1-400
bash $R/tools/fetch_calib.sh > /dev/null 2>&1; cp $R/gpurun_out/fetch_calib.txt $R/gpurun_out/profiles_new/r05_fetch_write_calibration.txt
