#!/bin/bash
# round 4, GPU call 34: every conv_fwd_kernel instance bounded to two blocks per CU (lab build -DCHAP_CONV_MINWAVES=2: 256 registers per lane) -- 2D and 3D step
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
L=$O/r04_conv_minwaves2_ab.log; : > $L
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
for rep in 1 2 3; do for v in "X=0" "CHAP_LIBPATH=tools/lab/libchap_hip_m2.so"; do for c in 2d 3d; do
  st=30; [ $c = 3d ] && st=20
  echo "== $c $v" >> $L
  env $v timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>$O/c34.err | ms >> $L || { tail -5 $O/c34.err; exit 1; }
done; done; done
echo "== stand-alone CHAP_LIBPATH=tools/lab/libchap_hip_m2.so" >> $L
CHAP_LIBPATH=tools/lab/libchap_hip_m2.so timeout -k 10 300 python3 tools/time_conv3d_rounds.py 2>/dev/null | grep conv3d >> $L
paste -d' ' - - < $L | head -14; tail -12 $L
