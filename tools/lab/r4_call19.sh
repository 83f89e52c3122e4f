#!/bin/bash
# round 4, GPU call 19: per-layer cost of the in-launch finalize against conv + bn_finalize (captured dependent chains)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/time_tail.py > $O/r04_tail_per_layer.log 2>&1; rc=$?
cat $O/r04_tail_per_layer.log; exit $rc
