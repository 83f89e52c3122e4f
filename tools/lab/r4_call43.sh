#!/bin/bash
# round 4, GPU call 43: the 3D weight-gradient bricks with a 256-register bound and an unpadded 32-wide B tile (80 960 B of LDS: two blocks per CU), 256 / 512 blocks
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 300 env CHAP_LIBPATH=tools/lab/libchap_hip_wg2.so python -m pytest tests/test_kernels_bwd_gpu.py -x -q -k "wgrad" 2>&1 | tail -2
L=$O/r04_wgrad3d_two_blocks_ab.log; : > $L
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('losses_finite'))"; }
for rep in 1 2 3; do for v in "X=0" "CHAP_LIBPATH=tools/lab/libchap_hip_wg2.so" "CHAP_LIBPATH=tools/lab/libchap_hip_wg2.so CHAP_WGRAD_BRICK_BLOCKS=512"; do
  echo "== 3d $v" >> $L
  env $v timeout -k 10 200 python3 bench.py --config 3d --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>$O/c43.err | ms >> $L || { tail -5 $O/c43.err; exit 1; }
done; done
for v in "X=0" "CHAP_LIBPATH=tools/lab/libchap_hip_wg2.so" "CHAP_LIBPATH=tools/lab/libchap_hip_wg2.so CHAP_WGRAD_BRICK_BLOCKS=512"; do
  echo "== shapes $v" >> $L
  env $v timeout -k 10 300 python3 tools/shape_table.py --config 3d --eager --reps 10 --only "wgrad    3D k3 s1 A=" 2>/dev/null | grep -E "^wgrad" | cut -c1-150 >> $L
done
cat $L
