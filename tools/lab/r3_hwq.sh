#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hq; mkdir -p $O; cd $R
b() { tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra $BARGS > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; return 0; }
  python3 -c "import json;d=json.load(open('$O/$tag.json'));print('%-14s %.3f ms %.1f vol/s'%('$tag', d['ms_per_step'], d['value']))"
}
for rep in 1 2 3 4; do
BARGS="--steps 30 --warmup 5"; b 2d_q4_$rep CHAP_X=0; b 2d_q8_$rep GPU_MAX_HW_QUEUES=8; b 2d_q6_$rep GPU_MAX_HW_QUEUES=6
BARGS="--config 3d --steps 20 --warmup 5"; b 3d_q4_$rep CHAP_X=0; b 3d_q8_$rep GPU_MAX_HW_QUEUES=8; b 3d_q6_$rep GPU_MAX_HW_QUEUES=6
done
