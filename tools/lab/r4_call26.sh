#!/bin/bash
# round 4, GPU call 26: steady-state timelines with the deferred decoder weight gradients (default now)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
for c in 2d 3d; do CHAP_LIBPATH=tools/lab/libchap_hip_lab.so timeout -k 10 300 python3 tools/timeline_untraced.py --config $c --out $O/r04_timeline_untraced_$c.json > $O/timeline_$c.log 2>&1 || { tail $O/timeline_$c.log; exit 1; }; tail -5 $O/timeline_$c.log | cut -c1-400; done
