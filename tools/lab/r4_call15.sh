#!/bin/bash
# round 4, GPU call 15: radix-select check, PMC traffic of a concat consumer with / without the dense halves, timelines, bench lines
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_kernels_bwd_gpu.py -x -q -k "lcc or diffmask or box" > $O/c15_tests.log 2>&1 || exit 1
cd /tmp; export TMPDIR=/tmp
for sc in 1 0; do for k in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_skip${sc}_$k
  CHAP_SPLIT_CONCAT=$sc rocprofv3 --pmc $k --output-format csv -d $O/pmc_skip${sc}_$k -- python3 $R/tools/shape_table.py --config 2d --eager --reps 5 --only "act_bwd C=16 @12x1x256x256 ng=2 +pool" > $O/pmc_skip${sc}_$k.log 2>&1
done
(cd $R && python3 tools/pmc_traffic.py --last 8 $O/pmc_skip${sc}_FETCH_SIZE $O/pmc_skip${sc}_WRITE_SIZE act_bwd > $O/r04_pmc_traffic_actbwd_skip_split${sc}.jsonl)
done
cd $R
for c in 2d 3d; do CHAP_LIBPATH=tools/lab/libchap_hip_lab.so python3 tools/timeline_untraced.py --config $c --out $O/r04_timeline_untraced_$c.json > $O/timeline_$c.log 2>&1 || exit 1; done
python3 bench.py > $O/r04_bench2d_bf16.json 2> $O/bench2d_bf16.err &&
python3 bench.py --config 3d --steps 20 --warmup 5 > $O/r04_bench3d_bf16.json 2> $O/bench3d_bf16.err
