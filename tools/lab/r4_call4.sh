#!/bin/bash
# round 4, GPU call 4: kernel-argument size A/B (CHAP_MAX_GROUP 4 / 2 / 1), timelines with the fixed lab build, V0-pass stage errors
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
B="python bench.py --no-cpu-baseline --no-extra --steps 40"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
{ for rep in 1 2; do for v in "X=0" "CHAP_LIBPATH=tools/lab/libchap_hip_g2.so" "CHAP_LIBPATH=tools/lab/libchap_hip_g1.so" "CHAP_GROUP=0" "CHAP_LIBPATH=tools/lab/libchap_hip_lab.so"; do
    echo "== 2d $v"; env $v timeout -k 10 200 $B 2>/dev/null | python -c "$J" || exit 1; done; done
  for v in "X=0" "CHAP_LIBPATH=tools/lab/libchap_hip_g2.so" "CHAP_LIBPATH=tools/lab/libchap_hip_g1.so" "CHAP_GROUP=0"; do
    echo "== 3d $v"; env $v timeout -k 10 200 $B --config 3d --steps 20 2>/dev/null | python -c "$J" || exit 1; done; } > $O/r4_kernarg_ab.log 2>&1 &&
CHAP_LIBPATH=tools/lab/libchap_hip_lab.so timeout -k 10 200 python tools/timeline_untraced.py --config 2d > $O/tl2d.log 2>&1 &&
CHAP_LIBPATH=tools/lab/libchap_hip_lab.so timeout -k 10 200 python tools/timeline_untraced.py --config 3d > $O/tl3d.log 2>&1 &&
CHAP_DIAG_SEEDS=6 timeout -k 10 400 python tests/diag_iteration_decisions.py > $O/diag_base2.log 2>&1
