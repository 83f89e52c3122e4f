#!/bin/bash
# Round 3: launch share (chap_set_launch_share) of pass B / the early VAT pass -- the passes with slack beside the iteration's long chain.
# HISTORICAL: the knobs CHAP_SHARE_B / CHAP_SHARE_PRE and the entry point existed only for this A/B (commit 0570834 removed them again: measured flat,
# profiles/r03_launch_share_ab.log); the script is kept as the record of what was run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sh; mkdir -p $O; cd $R
b() { tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra $BARGS > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -3 $O/$tag.err; return 0; }
  python3 - $tag $O/$tag.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); print("%-32s %.3f ms  %.1f vol/s"%(sys.argv[1], d["ms_per_step"], d["value"]))
P
}
for cfg in 2d 3d; do
  if [ $cfg = 2d ]; then BARGS="--steps 30 --warmup 5"; else BARGS="--config 3d --steps 20 --warmup 5"; fi
  b ${cfg}_base CHAP_X=0
  for sb in 75 50 35 25; do b ${cfg}_b$sb CHAP_SHARE_B=$sb; done
  b ${cfg}_pre50 CHAP_SHARE_PRE=50
  b ${cfg}_b50_pre50 CHAP_SHARE_B=50 CHAP_SHARE_PRE=50
  b ${cfg}_base2 CHAP_X=0
done
