#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/time_conv3d_rounds.py > $O/r04_conv3d_rounds.log 2>&1; rc=$?; cat $O/r04_conv3d_rounds.log; exit $rc
