#!/bin/bash
# round 4, GPU call 35: 2D (and 3D again) step with the three builds: product, register cap (m2), register cap + single staged weight buffer (wst1m2)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
L=$O/r04_conv_wst1m2_steps.log; : > $L
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
for rep in 1 2 3; do for v in "X=0" "CHAP_LIBPATH=tools/lab/libchap_hip_m2.so" "CHAP_LIBPATH=tools/lab/libchap_hip_wst1m2.so"; do for c in 2d 3d; do
  st=30; [ $c = 3d ] && st=20
  echo "== $c $v" >> $L
  env $v timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>$O/c35.err | ms >> $L || { tail -5 $O/c35.err; exit 1; }
done; done; done
paste -d' ' - - < $L
