#!/bin/bash
# round 4, GPU call 20: the direct first-layer kernel (conv_c1_mfma.h): tests, stand-alone timing, whole-iteration A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_bwd_gpu.py -x -q -k "first_conv" > $O/c20_tests.log 2>&1 || { tail -30 $O/c20_tests.log; exit 1; }
tail -2 $O/c20_tests.log
L=$O/r04_first_conv_ab.log
timeout -k 10 300 python3 tools/time_conv.py first > $L 2>&1 || { cat $L; exit 1; }
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
for rep in 1 2 3; do for t in 1 0; do for c in 2d 3d; do
  st=30; [ $c = 3d ] && st=20
  echo "== $c CHAP_C1_DIRECT=$t" >> $L
  CHAP_C1_DIRECT=$t timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>$O/c20_bench.err | ms >> $L || { tail -20 $O/c20_bench.err; exit 1; }
done; done; done
cat $L
