#!/bin/bash
# round 4, GPU call 36: the full GPU suite and the evidence run on the tree with two conv blocks per CU (256-register bound, one staged weight buffer)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > $O/r04_gpu_suite.log 2>&1; rc=$?
tail -18 $O/r04_gpu_suite.log
[ $rc = 0 ] || exit $rc
bash tools/lab/collect_r04.sh bench stats timeline > $O/collect.log 2>&1
tail -3 $O/collect.log
cut -c1-400 $O/r04_bench2d_bf16.json; echo
cut -c1-300 $O/r04_bench3d_bf16.json
