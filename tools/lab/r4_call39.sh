#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
CHAP_FULL_FP64=1 timeout -k 10 600 python -m pytest tests/test_iteration_conditioning_gpu.py -x -q -k "config0_2d or config1_2d" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-500
tail -2 gpurun_out/r04_iteration_parity.jsonl | cut -c1-1500
