#!/bin/bash
# round 4, GPU call 24: the decoders' weight gradients deferred to the forked stream (CHAP_DEFER_WGRAD): bitwise checks, then whole-iteration A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
CHAP_DEFER_WGRAD=1 timeout -k 10 600 python -m pytest tests/test_train_step_gpu.py -x -q > $O/c24_tests.log 2>&1 || { tail -30 $O/c24_tests.log; exit 1; }
tail -2 $O/c24_tests.log
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
L=$O/r04_defer_wgrad_ab.log; : > $L
for rep in 1 2 3; do for t in 1 0; do for c in 2d 3d; do
  st=30; [ $c = 3d ] && st=20
  echo "== $c CHAP_DEFER_WGRAD=$t" >> $L
  CHAP_DEFER_WGRAD=$t timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>$O/c24_bench.err | ms >> $L || { tail -20 $O/c24_bench.err; exit 1; }
done; done; done
paste -d' ' - - < $L
