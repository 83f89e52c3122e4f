#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_nt.log; : > $O
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name" >> $O
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 10 "$@" 2>>gpurun_out/r3_nt.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['losses_finite'])" >> $O
}
run 3d_base X=1 -- --config 3d
run 3d_nt1_minc64 CHAP_CONV_NT=1 CHAP_CONV_MINC=64 -- --config 3d
run 3d_nt1_minc32 CHAP_CONV_NT=1 CHAP_CONV_MINC=32 -- --config 3d
run 3d_nt1_minc64_nokpar CHAP_CONV_NT=1 CHAP_CONV_MINC=64 CHAP_CONV_KPAR=0 -- --config 3d
run 3d_occ2 CHAP_CONV_OCC_CAP=2 -- --config 3d
run 3d_base_again X=1 -- --config 3d
run 2d_base X=1 --
cat $O
