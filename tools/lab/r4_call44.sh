#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
L=$O/r04_wgrad_two_blocks_2d_ab.log; : > $L
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
for rep in 1 2 3; do for v in "X=0" "CHAP_LIBPATH=tools/lab/libchap_hip_wg2.so"; do
  echo "== 2d $v" >> $L
  env $v timeout -k 10 200 python3 bench.py --config 2d --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>$O/c44.err | ms >> $L || { tail -5 $O/c44.err; exit 1; }
done; done
paste -d' ' - - < $L
