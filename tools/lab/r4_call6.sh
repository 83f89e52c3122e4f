#!/bin/bash
# round 4, GPU call 6: the issue-order switches one by one; kpar2d inside the iteration
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
B="python bench.py --no-cpu-baseline --no-extra --steps 40"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
{ for v in "CHAP_ISSUE_INTERLEAVE=0" "CHAP_ISSUE_INTERLEAVE=1" "CHAP_ISSUE_INTERLEAVE=2" "CHAP_ISSUE_INTERLEAVE=4" "CHAP_ISSUE_INTERLEAVE=3" "CHAP_ISSUE_INTERLEAVE=0 CHAP_CONV_KPAR=1" "CHAP_ISSUE_INTERLEAVE=0"; do
    echo "== 2d $v"; env $v timeout -k 10 200 $B 2>/dev/null | python -c "$J" || exit 1; done
  for v in "CHAP_ISSUE_INTERLEAVE=0" "CHAP_ISSUE_INTERLEAVE=1" "CHAP_ISSUE_INTERLEAVE=2" "CHAP_ISSUE_INTERLEAVE=4"; do
    echo "== 3d $v"; env $v timeout -k 10 200 $B --config 3d --steps 20 2>/dev/null | python -c "$J" || exit 1; done; } > $O/r4_issue_ab2.log 2>&1
