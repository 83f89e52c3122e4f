#!/bin/bash
# round 3, last sweep: the launch-geometry knobs on the final schedule (grouped decoders in pass B / the early VAT pass)
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_knobs.log; : > $O
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo -n "$name " >> $O
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 10 "$@" 2>>gpurun_out/r3_knobs.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $O
}
run 2d_base X=1 --
run 2d_actbwd384 CHAP_ACTBWD_BLOCKS=384 --
run 2d_actbwd768 CHAP_ACTBWD_BLOCKS=768 --
run 2d_wg_512_256_256 CHAP_WGRAD_TARGETS=512,256,256 --
run 2d_wg_768_512_256 CHAP_WGRAD_TARGETS=768,512,256 --
run 2d_wg_1024_256_256 CHAP_WGRAD_TARGETS=1024,256,256 --
run 2d_occ2 CHAP_CONV_OCC_CAP=2 --
run 2d_grid50 CHAP_GRID_SCALE=50 --
run 2d_grid200 CHAP_GRID_SCALE=200 --
run 2d_wlds80 CHAP_CONV_WLDS_KB=80 --
run 2d_base2 X=1 --
run 3d_base X=1 -- --config 3d
run 3d_actbwd384 CHAP_ACTBWD_BLOCKS=384 -- --config 3d
run 3d_actbwd768 CHAP_ACTBWD_BLOCKS=768 -- --config 3d
run 3d_brickblocks384 CHAP_WGRAD_BRICK_BLOCKS=384 -- --config 3d
run 3d_kparmax400 CHAP_CONV_KPAR_MAX=400 -- --config 3d
run 3d_kparmax1600 CHAP_CONV_KPAR_MAX=1600 -- --config 3d
run 3d_grid50 CHAP_GRID_SCALE=50 -- --config 3d
run 3d_grid200 CHAP_GRID_SCALE=200 -- --config 3d
run 3d_base2 X=1 -- --config 3d
cat $O
