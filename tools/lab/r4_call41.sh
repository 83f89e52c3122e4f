#!/bin/bash
# round 4, GPU call 41: which of the two changes moves the fp32 iteration (config 0 loss 2.2e-6 -> 1.16e-5 when applied to the fp32 instances)?
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
for v in "X=0" "CHAP_LIBPATH=tools/lab/libchap_hip_f32minw2.so" "CHAP_LIBPATH=tools/lab/libchap_hip_f32wbuf1.so"; do
  echo "== $v"
  env $v timeout -k 10 300 python -m pytest tests/test_iteration_conditioning_gpu.py -x -q -k "config0_2d" 2>&1 | grep -E "AssertionError: \{|passed|failed" | cut -c1-300
  tail -1 gpurun_out/r04_iteration_parity.jsonl | cut -c150-420
done
