#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
(time timeout 1500 python -m pytest tests/test_downsample.py tests/test_gpu_parity_fullsize.py tests/test_gpu_profiler.py tests/test_gpu_registration.py tests/test_gpu_sharded.py -m gpu -x -q --durations=6) 2>&1 | tail -14
