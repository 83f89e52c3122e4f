#!/bin/bash
# round 4, GPU call 23: shape tables of the final tree; the default bench line once more with the refreshed PMC entry (roofline.traffic)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
bash tools/lab/collect_r04.sh shapes > $O/collect_shapes.log 2>&1; tail -2 $O/collect_shapes.log
python3 bench.py > $O/r04_bench2d_bf16.json 2> $O/bench2d_bf16.err; cut -c1-900 $O/r04_bench2d_bf16.json
