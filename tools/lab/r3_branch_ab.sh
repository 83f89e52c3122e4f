#!/bin/bash
# Round 3: decoders as parallel graph branches on one stream (CHAP_GROUP=3) and weight gradients as graph leaves (CHAP_WGRAD_LEAF=1):
# bitwise equality with the eager iteration first, then whole-iteration A/B on this box.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/br; mkdir -p $O; cd $R
if [ "$1" != "noparity" ]; then
CHAP_GROUP=3 CHAP_WGRAD_LEAF=1 timeout -k 10 600 python3 -m pytest tests/test_train_step_gpu.py -x -q > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
fi
b() { # tag, env..., -- bench args
  tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra $BARGS > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; return 1; }
  python3 - $tag $O/$tag.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); print("%-24s %.3f ms  %.1f vol/s"%(sys.argv[1], d["ms_per_step"], d["value"]))
P
}
for rep in 1 2; do
BARGS="--steps 30 --warmup 5"
b 2d_base_$rep CHAP_GROUP=1 && b 2d_g3_$rep CHAP_GROUP=3 && b 2d_leaf_$rep CHAP_GROUP=1 CHAP_WGRAD_LEAF=1 && b 2d_g3leaf_$rep CHAP_GROUP=3 CHAP_WGRAD_LEAF=1 || exit 1
BARGS="--config 3d --steps 20 --warmup 5"
b 3d_base_$rep CHAP_GROUP=1 && b 3d_g3_$rep CHAP_GROUP=3 && b 3d_leaf_$rep CHAP_GROUP=1 CHAP_WGRAD_LEAF=1 && b 3d_g3leaf_$rep CHAP_GROUP=3 CHAP_WGRAD_LEAF=1 || exit 1
done
