#!/bin/bash
# round 4, GPU call 32: the 3D 64-channel bricks with a SINGLE staged weight buffer (lab build -DCHAP_CONV_WST1: 69 KB of LDS, two blocks per CU) against the
# resident-weights default (153 KB, one block per CU) and the double-buffered staging (97 KB)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
L=$O/r04_conv3d_wst1_ab.log; : > $L
for v in "X=0" "CHAP_CONV_WLDS_KB=100" "CHAP_CONV_WLDS_KB=100 CHAP_LIBPATH=tools/lab/libchap_hip_wst1.so"; do
  echo "== stand-alone $v" >> $L
  env $v timeout -k 10 300 python3 tools/time_conv3d_rounds.py 2>/dev/null | grep conv3d >> $L || exit 1
done
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
for rep in 1 2; do for v in "X=0" "CHAP_CONV_WLDS_KB=100" "CHAP_CONV_WLDS_KB=100 CHAP_LIBPATH=tools/lab/libchap_hip_wst1.so" "CHAP_LIBPATH=tools/lab/libchap_hip_wst1.so"; do
  echo "== 3d step $v" >> $L
  env $v timeout -k 10 200 python3 bench.py --config 3d --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>$O/c32.err | ms >> $L || { tail -5 $O/c32.err; exit 1; }
done; done
cat $L
