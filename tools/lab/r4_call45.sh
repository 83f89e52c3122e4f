#!/bin/bash
# round 4, GPU call 45: after removing the issue-order switches: iteration tests (bitwise eager == graph, oracle parity, DP), bench lines
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_train_step_gpu.py tests/test_parallel_gpu.py tests/test_parity_gates_gpu.py -x -q > $O/c45_tests.log 2>&1; rc=$?
tail -3 $O/c45_tests.log; [ $rc = 0 ] || exit $rc
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for c in 2d 3d; do st=30; [ $c = 3d ] && st=20; echo "== $c"; timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | ms; done
