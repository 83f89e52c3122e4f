#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lf; mkdir -p $O; cd $R
b() { tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra $BARGS > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; return 1; }
  python3 - $tag $O/$tag.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); print("%-28s %.3f ms  %.1f vol/s"%(sys.argv[1], d["ms_per_step"], d["value"]))
P
}
for rep in 1 2; do
BARGS="--steps 30 --warmup 5"
b 2d_base_$rep CHAP_GROUP=1 && b 2d_leaf2_$rep CHAP_WGRAD_LEAF=2 || exit 1
BARGS="--config 3d --steps 20 --warmup 5"
b 3d_base_$rep CHAP_GROUP=1 && b 3d_leaf2_$rep CHAP_WGRAD_LEAF=2 || exit 1
done
python3 tools/ablate_step.py > $O/ablate2d.log 2>&1; tail -12 $O/ablate2d.log
