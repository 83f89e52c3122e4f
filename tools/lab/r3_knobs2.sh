#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kn2; mkdir -p $O; cd $R
b() { tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra $BARGS > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -3 $O/$tag.err; return 0; }
  python3 - $tag $O/$tag.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); print("%-32s %.3f ms  %.1f vol/s"%(sys.argv[1], d["ms_per_step"], d["value"]))
P
}
BARGS="--steps 30 --warmup 5"
b 2d_base CHAP_X=0; b 2d_kpar1 CHAP_CONV_KPAR=1; b 2d_kpar0 CHAP_CONV_KPAR=0; b 2d_base2 CHAP_X=0; b 2d_kpar1b CHAP_CONV_KPAR=1
BARGS="--config 3d --steps 20 --warmup 5"
b 3d_base CHAP_X=0; b 3d_occ1 CHAP_CONV_OCC_CAP=1; b 3d_occ2 CHAP_CONV_OCC_CAP=2; b 3d_kpar0 CHAP_CONV_KPAR=0; b 3d_kpar1 CHAP_CONV_KPAR=1; b 3d_base2 CHAP_X=0
