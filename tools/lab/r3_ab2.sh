#!/bin/bash
# round 3: A/B of the folded BatchNorm-backward totals and of the host-known statistics slot count
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r3_ab2.log
: > $O
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name" >> $O
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 10 "$@" 2>>gpurun_out/r3_ab2.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['losses_finite'])" >> $O
}
for i in 1 2; do
for c in "2d --" "3d -- --config 3d"; do
set -- $c; n=$1; shift
run ${n}_base X=1 "$@"
run ${n}_nofold CHAP_ACTBWD_FOLD=0 "$@"
run ${n}_noknown CHAP_BNFIN_KNOWN_SLOTS=0 "$@"
run ${n}_neither CHAP_ACTBWD_FOLD=0 CHAP_BNFIN_KNOWN_SLOTS=0 "$@"
done
done
cat $O
