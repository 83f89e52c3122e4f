#!/bin/bash
# 2D: the K-parallel kernel for the deep layers (CHAP_CONV_KPAR=1) against the default, three A/B pairs + the stand-alone shape times
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kp; mkdir -p $O; cd $R
b() { tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; return 0; }
  python3 - $tag $O/$tag.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); print("%-16s %.3f ms  %.1f vol/s   top_kernel %.2f us frac %.4f"%(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["top_kernel"]["avg_launch_us"], d["roofline"]["top_kernel"]["frac"]))
P
}
for rep in 1 2 3; do b base_$rep CHAP_X=0; b kpar_$rep CHAP_CONV_KPAR=1; done
for k in 2 1; do
  CHAP_CONV_KPAR=$k python3 tools/shape_table.py --config 2d --only "conv_fwd 2D k3 s1" --out $O/s_$k.csv > $O/s_$k.log 2>&1
  awk -F, -v k=$k 'NR>1 && ($2 ~ /128|256|64->64/) {printf "kpar=%s %s us=%s TF=%s | %s\n",k,$2,$4,$10,substr($13,1,40)}' $O/s_$k.csv
done
