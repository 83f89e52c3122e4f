#!/bin/bash
# round 4, GPU call 7: issue-order switches under other queue counts; timelines of the single switches
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
B="python bench.py --no-cpu-baseline --no-extra --steps 40"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
CHAP_ISSUE_INTERLEAVE=1 CHAP_LIBPATH=tools/lab/libchap_hip_lab.so timeout -k 10 200 python tools/timeline_untraced.py --config 2d --out $O/r04_timeline_untraced_2d_i1.json > $O/tl2d_i1.log 2>&1 &&
CHAP_ISSUE_INTERLEAVE=2 CHAP_LIBPATH=tools/lab/libchap_hip_lab.so timeout -k 10 200 python tools/timeline_untraced.py --config 2d --out $O/r04_timeline_untraced_2d_i2.json > $O/tl2d_i2.log 2>&1 &&
{ for q in "X=0" "GPU_MAX_HW_QUEUES=5" "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_FORCE_GRAPH_QUEUES=6" "DEBUG_HIP_FORCE_GRAPH_QUEUES=3"; do for i in 0 1 2 3; do
    echo "== 2d $q CHAP_ISSUE_INTERLEAVE=$i"; env $q CHAP_ISSUE_INTERLEAVE=$i timeout -k 10 200 $B 2>/dev/null | python -c "$J" || echo failed; done; done; } > $O/r4_issue_ab3.log 2>&1
