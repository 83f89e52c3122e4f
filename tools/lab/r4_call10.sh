#!/bin/bash
# round 4, GPU call 10: wave-private weight gradient on the deep 2D layers; fork mask 14 vs 15
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
B="python bench.py --no-cpu-baseline --no-extra --steps 40"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
timeout -k 10 300 python -m pytest tests/test_kernels_bwd_gpu.py -x -q -k wgrad > $O/r4_c10_tests.log 2>&1 &&
{ for v in "CHAP_WGRAD_WP=1024" "CHAP_WGRAD_WP=1"; do echo "== $v"; env $v timeout -k 10 120 python tools/time_wgrad2d.py || exit 1; done; } > $O/r4_wp_deep_time.log 2>&1 &&
{ for rep in 1 2 3; do for v in "CHAP_WGRAD_WP=1024" "CHAP_WGRAD_WP=256" "CHAP_WGRAD_WP=64" "CHAP_WGRAD_WP=1" "CHAP_FORK_MASK=14"; do
    echo "== 2d $v"; env $v timeout -k 10 200 $B 2>/dev/null | python -c "$J" || exit 1; done; done
  for rep in 1 2; do for v in "CHAP_FORK_MASK=15" "CHAP_FORK_MASK=14"; do echo "== 3d $v"; env $v timeout -k 10 200 $B --config 3d --steps 20 2>/dev/null | python -c "$J" || exit 1; done; done; } > $O/r4_wp_deep_bench.log 2>&1
