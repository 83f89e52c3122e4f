#!/bin/bash
# Concurrency timeline of the 2D / 3D iteration (tools/timeline.py): who owns the step time outright, where the chain idles.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in ${*:-2d 3d}; do
  rm -rf $O/tr$c
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$c -- python3 $R/bench.py --config $c --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $O/tr$c.json 2> $O/tr$c.err
  python3 $R/tools/timeline.py $O/tr$c --iters 8 --top 40 > $O/timeline_$c.json
  echo "timeline $c done"
done
