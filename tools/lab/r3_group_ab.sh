#!/bin/bash
# round 3: A/B of grouped launches. CHAP_GROUP=0: never (round 2), 1: in the passes that cannot fork a second stream, 2: everywhere
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r3_group_ab.log
: > $O
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name" >> $O
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 10 "$@" 2>>gpurun_out/r3_group_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['losses_finite'])" >> $O
}
for i in 1 2; do
for g in 0 1 2; do
run 2d_group$g CHAP_GROUP=$g --
run 3d_group$g CHAP_GROUP=$g -- --config 3d
done
done
cat $O
