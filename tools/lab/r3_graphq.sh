#!/bin/bash
# Round 3: how many internal streams the HIP graph executor uses (DEBUG_HIP_FORCE_GRAPH_QUEUES) x the graph shapes of tools/lab/r3_branch_ab.sh
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gq; mkdir -p $O; cd $R
b() { tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra $BARGS > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; return 1; }
  python3 - $tag $O/$tag.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); print("%-28s %.3f ms  %.1f vol/s"%(sys.argv[1], d["ms_per_step"], d["value"]))
P
}
for cfg in 2d 3d; do
  if [ $cfg = 2d ]; then BARGS="--steps 30 --warmup 5"; else BARGS="--config 3d --steps 20 --warmup 5"; fi
  for q in default 1 2 3 6 8 16; do
    if [ $q = default ]; then QE="CHAP_X=0"; else QE="DEBUG_HIP_FORCE_GRAPH_QUEUES=$q"; fi
    b ${cfg}_base_q$q CHAP_GROUP=1 $QE || exit 1
    b ${cfg}_g3_q$q CHAP_GROUP=3 $QE || exit 1
    b ${cfg}_leaf_q$q CHAP_GROUP=1 CHAP_WGRAD_LEAF=1 $QE || exit 1
    b ${cfg}_g3leaf_q$q CHAP_GROUP=3 CHAP_WGRAD_LEAF=1 $QE || exit 1
  done
done
