#!/bin/bash
# round 3: XCD-aware weight-gradient tile assignment (now the default) + the upper bound of the "lazy gradient" fusion
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 $R/bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 10 "$@" 2>>$O/err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'], d['config']['losses_finite'])"
}
for i in 1 2; do
run 2d_default X=1 --
run 3d_default X=1 -- --config 3d
run 2d_skip_actapply CHAP_LAB_SKIP_ACTAPPLY=1 --
run 3d_skip_actapply CHAP_LAB_SKIP_ACTAPPLY=1 -- --config 3d
done
for k in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_wgrad_$k
  rocprofv3 --pmc $k --output-format csv -d $O/pmc_wgrad_$k -- python3 $R/tools/shape_table.py --config 2d --eager --reps 5 --only "wgrad 2D k3 s1 A=16 B=16 @256x256 N=12 A=ar" > $O/pmc_wgrad_$k.log 2>&1
done
(cd $R && python3 tools/pmc_traffic.py --last 8 $O/pmc_wgrad_FETCH_SIZE $O/pmc_wgrad_WRITE_SIZE wgrad_kernel | tee $O/r03_pmc_traffic_wgrad_after.jsonl | cut -c1-300)
grep "wgrad 2D k3 s1 A=16 B=16 @256x256" $O/pmc_wgrad_FETCH_SIZE.log | tail -3
find $O -name "*counter_collection.csv" -size +4M -delete
