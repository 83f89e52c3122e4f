#!/bin/bash
# round 4, GPU call 31: smoke() and the full GPU suite with durations on the committed tree
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/c31_smoke.log 2>&1 || { tail -20 $O/c31_smoke.log; exit 1; }
tail -2 $O/c31_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=25 > $O/r04_gpu_suite.log 2>&1; rc=$?
tail -40 $O/r04_gpu_suite.log
exit $rc
