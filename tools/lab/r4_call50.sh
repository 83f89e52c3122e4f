#!/bin/bash
# round 4, GPU call 50: shape tables of the round's last tree
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
bash tools/lab/collect_r04.sh shapes > $O/collect_shapes.log 2>&1; tail -2 $O/collect_shapes.log; wc -l $O/r04_conv_shapes_2d.csv $O/r04_conv_shapes_3d.csv
