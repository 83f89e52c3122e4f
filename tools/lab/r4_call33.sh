#!/bin/bash
# round 4, GPU call 33: 3D bricks with a single staged weight buffer AND a 256-register cap (lab build -DCHAP_CONV_WST1 -DCHAP_CONV_MINWAVES=2): two blocks per CU
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
L=$O/r04_conv3d_wst1m2_ab.log; : > $L
for v in "X=0" "CHAP_CONV_WLDS_KB=100 CHAP_LIBPATH=tools/lab/libchap_hip_wst1m2.so" "CHAP_CONV_WLDS_KB=60 CHAP_LIBPATH=tools/lab/libchap_hip_wst1m2.so"; do
  echo "== stand-alone $v" >> $L
  env $v timeout -k 10 300 python3 tools/time_conv3d_rounds.py 2>/dev/null | grep conv3d >> $L || exit 1
done
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
for rep in 1 2; do for v in "X=0" "CHAP_CONV_WLDS_KB=100 CHAP_LIBPATH=tools/lab/libchap_hip_wst1m2.so" "CHAP_CONV_WLDS_KB=60 CHAP_LIBPATH=tools/lab/libchap_hip_wst1m2.so" "CHAP_LIBPATH=tools/lab/libchap_hip_wst1m2.so"; do
  echo "== 3d step $v" >> $L
  env $v timeout -k 10 200 python3 bench.py --config 3d --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>$O/c33.err | ms >> $L || { tail -5 $O/c33.err; exit 1; }
done; done
cat $L
