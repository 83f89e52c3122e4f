#!/bin/bash
# round 4, GPU call 29: deferred weight gradients -- the forked decoder's own ones right behind its chain (CHAP_DEFER_OWN_FIRST) x which decoder is forked (CHAP_SIDE_DECODER)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
CHAP_DEFER_OWN_FIRST=1 timeout -k 10 600 python -m pytest tests/test_train_step_gpu.py -x -q > $O/c29_tests.log 2>&1 || { tail -30 $O/c29_tests.log; exit 1; }
tail -2 $O/c29_tests.log
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
L=$O/r04_defer_own_first_ab.log; : > $L
for rep in 1 2; do for of in 0 1; do for sd in 1 2; do for c in 2d 3d; do
  st=30; [ $c = 3d ] && st=20
  echo "== $c CHAP_DEFER_OWN_FIRST=$of CHAP_SIDE_DECODER=$sd" >> $L
  CHAP_DEFER_OWN_FIRST=$of CHAP_SIDE_DECODER=$sd timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>$O/c29_bench.err | ms >> $L || { tail -20 $O/c29_bench.err; exit 1; }
done; done; done; done
paste -d' ' - - < $L
