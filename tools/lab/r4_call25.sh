#!/bin/bash
# round 4, GPU call 25: CHAP_DEFER_WGRAD 0 / 1 / 2 (2: the trunk's weight gradients on the forked stream too)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
CHAP_DEFER_WGRAD=2 timeout -k 10 600 python -m pytest tests/test_train_step_gpu.py -x -q > $O/c25_tests.log 2>&1 || { tail -30 $O/c25_tests.log; exit 1; }
tail -2 $O/c25_tests.log
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
L=$O/r04_defer_wgrad_ab2.log; : > $L
for rep in 1 2 3; do for t in 2 1 0; do for c in 2d 3d; do
  st=30; [ $c = 3d ] && st=20
  echo "== $c CHAP_DEFER_WGRAD=$t" >> $L
  CHAP_DEFER_WGRAD=$t timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>$O/c25_bench.err | ms >> $L || { tail -20 $O/c25_bench.err; exit 1; }
done; done; done
paste -d' ' - - < $L
