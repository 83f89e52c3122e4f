#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_iteration_conditioning_gpu.py -x -q -k "config0_2d or config1_2d or config3" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-500
tail -3 gpurun_out/r04_iteration_parity.jsonl | cut -c1-520
