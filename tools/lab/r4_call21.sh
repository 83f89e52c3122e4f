#!/bin/bash
# round 4, GPU call 21: blocking of the mid-deep 3D layers (64 ch at 28x28x20, 128 at 14x14x10) -- more, smaller blocks against the latency of a one-block-per-CU grid
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
L=$O/r04_conv3d_blocking_ab.log; : > $L
run() { lab=$1; shift; echo "== 3d $lab" >> $L; env "$@" timeout -k 10 200 python3 bench.py --config 3d --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>$O/c21.err | ms >> $L || { tail -5 $O/c21.err; exit 1; }; }
for rep in 1 2; do
run "default" X=0 || exit 1
run "NT=1 for Cin>=64" CHAP_CONV_NT=1 || exit 1
run "NT=1 for Cin>=32" CHAP_CONV_NT=1 CHAP_CONV_MINC=32 || exit 1
run "MR=2 for Cin>=64" CHAP_CONV_MR=2 || exit 1
run "MR=1 for Cin>=64" CHAP_CONV_MR=1 || exit 1
run "kpar up to 1600 blocks" CHAP_CONV_KPAR_MAX=1600 || exit 1
done
for v in "X=0" "CHAP_CONV_NT=1" "CHAP_CONV_MR=2" "CHAP_CONV_MR=1"; do
  echo "== shapes $v" >> $L
  env $v timeout -k 10 300 python3 tools/shape_table.py --config 3d --eager --reps 10 --only "conv_fwd 3D k3 s1 64->64 @28x28x20" 2>/dev/null | grep -E "conv_fwd" | cut -c1-160 >> $L
done
cat $L
