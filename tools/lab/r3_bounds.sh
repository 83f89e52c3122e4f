#!/bin/bash
# round 3: upper bounds for taking bn_finalize / act_bwd_sum off the launch chain (timing only: the skip knobs break the numerics)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r3_bounds.log
: > $O
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name" >> $O
  env "${envs[@]}" python bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>>gpurun_out/r3_bounds.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['losses_finite'])" >> $O
}
run 2d_base X=1 --
run 2d_skip_bnfin CHAP_LAB_SKIP_BNFIN=1 --
run 2d_skip_actsum CHAP_LAB_SKIP_ACTSUM=1 --
run 2d_skip_both CHAP_LAB_SKIP_BNFIN=1 CHAP_LAB_SKIP_ACTSUM=1 --
run 2d_occ1 CHAP_CONV_OCC_CAP=1 --
run 2d_occ2 CHAP_CONV_OCC_CAP=2 --
run 2d_actbwd256 CHAP_ACTBWD_BLOCKS=256 --
run 2d_base_again X=1 --
run 3d_base X=1 -- --config 3d
run 3d_skip_bnfin CHAP_LAB_SKIP_BNFIN=1 -- --config 3d
run 3d_skip_actsum CHAP_LAB_SKIP_ACTSUM=1 -- --config 3d
run 3d_skip_both CHAP_LAB_SKIP_BNFIN=1 CHAP_LAB_SKIP_ACTSUM=1 -- --config 3d
run 3d_occ1 CHAP_CONV_OCC_CAP=1 -- --config 3d
cat $O
