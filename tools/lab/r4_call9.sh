#!/bin/bash
# round 4, GPU call 9: issue order x fork mask (which chains collide on the executor's queues)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
B="python bench.py --no-cpu-baseline --no-extra --steps 40"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
{ for i in 1 2 3; do for m in 15 14 13 11 7 12 10 6; do echo "== 2d CHAP_ISSUE_INTERLEAVE=$i CHAP_FORK_MASK=$m"; CHAP_ISSUE_INTERLEAVE=$i CHAP_FORK_MASK=$m timeout -k 10 200 $B 2>/dev/null | python -c "$J" || echo failed; done; done; } > $O/r4_issue_fork.log 2>&1
