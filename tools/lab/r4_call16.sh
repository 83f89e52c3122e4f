#!/bin/bash
# round 4, GPU call 16: which decoder rides the forked stream (CHAP_SIDE_DECODER) A/B, PMC traffic of the 2D roofline kernel (now conv_wp_kernel)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
L=$O/r04_side_decoder_ab.log; : > $L
for rep in 1 2; do for sd in 2 1; do
  echo "== CHAP_SIDE_DECODER=$sd 2d" >> $L
  CHAP_SIDE_DECODER=$sd timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $L || exit 1
  echo "== CHAP_SIDE_DECODER=$sd 3d" >> $L
  CHAP_SIDE_DECODER=$sd timeout -k 10 200 python3 bench.py --config 3d --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $L || exit 1
done; done
cd /tmp; export TMPDIR=/tmp
for k in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_dom2d_$k
  rocprofv3 --pmc $k --output-format csv -d $O/pmc_dom2d_$k -- python3 $R/tools/dominant_kernel.py 12 > $O/pmc_dom2d_$k.log 2>&1
done
(cd $R && python3 tools/pmc_traffic.py $O/pmc_dom2d_FETCH_SIZE $O/pmc_dom2d_WRITE_SIZE conv_ > $O/r04_pmc_traffic_dominant_2d.jsonl)
