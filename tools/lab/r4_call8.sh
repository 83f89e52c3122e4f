#!/bin/bash
# round 4, GPU call 8: out2 test + A/B, kpar2d trace and in-iteration A/B, fork-mask sweep
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
B="python bench.py --no-cpu-baseline --no-extra --steps 40"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "two_dense or k_parallel" > $O/r4_c8_tests.log 2>&1 &&
timeout -k 10 300 python -m pytest tests/test_train_step_gpu.py -x -q > $O/r4_c8_tests2.log 2>&1 &&
{ for rep in 1 2 3; do for v in "X=0" "CHAP_SPLIT_CONCAT=0" "CHAP_CONV_KPAR=1"; do
    echo "== 2d $v"; env $v timeout -k 10 200 $B 2>/dev/null | python -c "$J" || exit 1; done; done; } > $O/r4_c8_ab.log 2>&1 &&
{ for m in 15 14 13 11 7 12 10 6 9 5 3 0; do echo "== 2d CHAP_FORK_MASK=$m"; CHAP_FORK_MASK=$m timeout -k 10 200 $B 2>/dev/null | python -c "$J" || echo failed; done
  for m in 15 14 13 11 7 0; do echo "== 3d CHAP_FORK_MASK=$m"; CHAP_FORK_MASK=$m timeout -k 10 200 $B --config 3d --steps 20 2>/dev/null | python -c "$J" || echo failed; done; } > $O/r4_forkmask.log 2>&1 &&
timeout -k 10 200 bash tools/lab/r3_convtrace.sh q > $O/ct_q.log 2>&1
