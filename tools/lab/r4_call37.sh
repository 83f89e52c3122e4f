#!/bin/bash
# round 4, GPU call 37: is the config-0 loss distance (1.16e-5 against the bound 5e-6) the statistics re-dealt over twice the blocks?  Same test with the persistent grids capped at one block per CU
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
for v in "X=0" "X=1" "CHAP_CONV_OCC_CAP=1"; do
  echo "== $v"
  env $v timeout -k 10 300 python -m pytest tests/test_iteration_conditioning_gpu.py -x -q -k "config0_2d" 2>&1 | grep -E "AssertionError: \{|passed|failed" | cut -c1-400
done
tail -3 gpurun_out/r04_iteration_parity.jsonl | cut -c1-600
