#!/bin/bash
# round 4, GPU call 18: in-launch totals (csrc/tail.h) -- kernel tests, whole-iteration A/B (CHAP_TAIL=0/1), then the full GPU suite
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_bwd_gpu.py -x -q -k "finalizes or in_launch or act_bn or conv3x3 or wave_private or deconv" > $O/c18_kernel_tests.log 2>&1 || { tail -30 $O/c18_kernel_tests.log; exit 1; }
tail -2 $O/c18_kernel_tests.log
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
L=$O/r04_tail_ab.log; : > $L
for rep in 1 2; do for t in 1 0; do for c in 2d 3d; do
  st=30; [ $c = 3d ] && st=20
  echo "== $c CHAP_TAIL=$t" >> $L
  CHAP_TAIL=$t timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>$O/c18_bench.err | ms >> $L || { tail -20 $O/c18_bench.err; exit 1; }
done; done; done
cat $L
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/c18_gpu_suite.log 2>&1; rc=$?
tail -5 $O/c18_gpu_suite.log
exit $rc
