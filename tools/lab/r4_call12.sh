#!/bin/bash
# round 4, GPU call 12: wave-private conv of the shallow 2D layers: test, stand-alone timings, whole-iteration A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
B="python bench.py --no-cpu-baseline --no-extra --steps 40"
J='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "wave_private or two_dense" > $O/r4_c12_tests.log 2>&1 &&
{ for v in "CHAP_CONV_WP=0" "CHAP_CONV_WP=1" "CHAP_CONV_WP=1 CHAP_CONV_WP_BPC=3" "CHAP_CONV_WP=1 CHAP_CONV_WP_BPC=2" "CHAP_CONV_WP=1 CHAP_CONV_WP_BPC=6"; do echo "== $v"; env $v timeout -k 10 120 python tools/time_conv.py shallow || exit 1; done; } > $O/r4_convwp_time.log 2>&1 &&
{ for rep in 1 2 3; do for v in "CHAP_CONV_WP=0" "CHAP_CONV_WP=1" "CHAP_CONV_WP=4096"; do
    echo "== 2d $v"; env $v timeout -k 10 200 $B 2>/dev/null | python -c "$J" || exit 1; done; done; } > $O/r4_convwp_bench.log 2>&1
