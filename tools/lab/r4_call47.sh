#!/bin/bash
# round 4, GPU call 47: the full GPU suite on the round's last tree
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > $O/r04_gpu_suite.log 2>&1; rc=$?
tail -16 $O/r04_gpu_suite.log
exit $rc
