#!/bin/bash
# round 3: 3D brick kernels, contiguous runs (default) vs round-robin tile walk (libchap_hip_walk.so = -DCHAP_ZW_INTERLEAVE=1)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03walk; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in base walk; do
  [ $v = walk ] && export CHAP_LIBPATH=$R/chap_amd/libchap_hip_walk.so || unset CHAP_LIBPATH
  for i in 1 2; do python3 $R/bench.py --config 3d --steps 30 --warmup 5 --no-cpu-baseline 2>$O/b_$v.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['top_kernel']['avg_launch_us'])"; done
  for k in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${v}_$k
    rocprofv3 --pmc $k --output-format csv -d $O/pmc_${v}_$k -- python3 $R/tools/dominant_kernel.py 3d > $O/pmc_${v}_$k.log 2>&1
  done
  (cd $R && python3 tools/pmc_traffic.py $O/pmc_${v}_FETCH_SIZE $O/pmc_${v}_WRITE_SIZE conv_fwd_kernel | tee $O/r03_pmc_traffic_walk_$v.jsonl)
done
find $O -name "*counter_collection.csv" -size +4M -delete
