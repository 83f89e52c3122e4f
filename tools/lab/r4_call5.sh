#!/bin/bash
# round 4, GPU call 5: issue-order interleave A/B (2D, 3D), correctness subset, kpar2d timings + A/B, timelines after
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
B="python bench.py --no-cpu-baseline --no-extra --steps 40"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
timeout -k 10 600 python -m pytest tests/test_train_step_gpu.py tests/test_kernels_gpu.py -x -q -k "not dice" > $O/r4_c5_tests.log 2>&1 &&
{ for rep in 1 2; do for v in "CHAP_ISSUE_INTERLEAVE=0" "CHAP_ISSUE_INTERLEAVE=1" "CHAP_ISSUE_INTERLEAVE=1 CHAP_CONV_KPAR=1"; do
    echo "== 2d $v"; env $v timeout -k 10 200 $B 2>/dev/null | python -c "$J" || exit 1; done; done
  for rep in 1 2; do for v in "CHAP_ISSUE_INTERLEAVE=0" "CHAP_ISSUE_INTERLEAVE=1"; do
    echo "== 3d $v"; env $v timeout -k 10 200 $B --config 3d --steps 20 2>/dev/null | python -c "$J" || exit 1; done; done; } > $O/r4_issue_ab.log 2>&1 &&
{ for v in "CHAP_CONV_KPAR=0" "CHAP_CONV_KPAR=1"; do echo "== $v"; env $v timeout -k 10 200 python tools/time_conv.py deep || exit 1; done; } > $O/r4_kpar2d_time.log 2>&1 &&
CHAP_LIBPATH=tools/lab/libchap_hip_lab.so timeout -k 10 200 python tools/timeline_untraced.py --config 2d > $O/tl2d.log 2>&1 &&
CHAP_LIBPATH=tools/lab/libchap_hip_lab.so timeout -k 10 200 python tools/timeline_untraced.py --config 3d > $O/tl3d.log 2>&1
