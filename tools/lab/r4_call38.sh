#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_bwd_gpu.py tests/test_train_step_gpu.py -q > $O/c38_tests.log 2>&1; rc=$?
grep -E "^FAILED|passed|failed" $O/c38_tests.log | head -30
exit 0
