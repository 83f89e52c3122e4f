#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bf; mkdir -p $O; cd $R
python3 -m pytest tests/test_kernels_gpu.py tests/test_net2d_gpu.py tests/test_net3d_gpu.py tests/test_train_step_gpu.py tests/test_parallel_gpu.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for k in 0 1; do for c in 2d 3d; do
  CHAP_BNFIN_COALESCED=$k python3 tools/shape_table.py --config $c --only "bn_finalize" --out $O/s_${c}_$k.csv > $O/s_${c}_$k.log 2>&1
  awk -F, -v k=$k 'NR>1{printf "coalesced=%s %s %s us=%s\n",k,$1,$2,$4}' $O/s_${c}_$k.csv
done; done
b() { tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra $BARGS > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; return 0; }
  python3 -c "import json;d=json.load(open('$O/$tag.json'));print('%-12s %.3f ms %.1f vol/s'%('$tag', d['ms_per_step'], d['value']))"
}
for rep in 1 2 3; do
BARGS="--steps 30 --warmup 5"; b 2d_old_$rep CHAP_BNFIN_COALESCED=0; b 2d_new_$rep CHAP_BNFIN_COALESCED=1
BARGS="--config 3d --steps 20 --warmup 5"; b 3d_old_$rep CHAP_BNFIN_COALESCED=0; b 3d_new_$rep CHAP_BNFIN_COALESCED=1
done
