#!/bin/bash
# round 4, GPU call 3: untraced timelines, kpar phase trace, wave-private wgrad tests + timings + whole-iteration A/B, fusion bounds
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
LAB=tools/lab/libchap_hip_lab.so
CHAP_LIBPATH=$LAB timeout -k 10 200 python tools/timeline_untraced.py --config 2d > $O/tl2d.log 2>&1 &&
CHAP_LIBPATH=$LAB timeout -k 10 200 python tools/timeline_untraced.py --config 3d > $O/tl3d.log 2>&1 &&
timeout -k 10 200 bash tools/lab/r3_convtrace.sh p > $O/ct_p.log 2>&1 &&
timeout -k 10 300 python -m pytest tests/test_kernels_bwd_gpu.py -x -q -k wgrad > $O/r4_wp_tests.log 2>&1 &&
{ for v in "CHAP_WGRAD_WP=0" "CHAP_WGRAD_WP=1024" "CHAP_WGRAD_WP=1024 CHAP_WGRAD_WP_MR=1" "CHAP_WGRAD_WP=1024 CHAP_WGRAD_WP_BLOCKS=768" "CHAP_WGRAD_WP=1024 CHAP_WGRAD_WP_BLOCKS=1024" "CHAP_WGRAD_WP=1024 CHAP_WGRAD_WP_BLOCKS=256"; do
    echo "== $v"; env $v timeout -k 10 120 python tools/time_wgrad2d.py || exit 1; done; } > $O/r4_wp_time.log 2>&1 &&
{ for rep in 1 2; do for v in "CHAP_WGRAD_WP=0" "CHAP_WGRAD_WP=1024" "CHAP_WGRAD_WP=1024 CHAP_WGRAD_WP_MR=1" "CHAP_WGRAD_WP=256"; do
    echo "== $v"; env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" || exit 1; done; done; } > $O/r4_wp_bench.log 2>&1 &&
{ for v in "X=0" "CHAP_LAB_SKIP_POOL=1"; do echo "== 2d $v"; env $v CHAP_LIBPATH=$LAB timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" || exit 1; done
  for v in "X=0" "CHAP_LAB_SKIP_UPSAMPLE=1" "CHAP_LAB_SKIP_UPSAMPLE=1 CHAP_LAB_SKIP_UPSAMPLE_BWD=1"; do echo "== 3d $v"; env $v CHAP_LIBPATH=$LAB timeout -k 10 200 python bench.py --config 3d --no-cpu-baseline --no-extra --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" || exit 1; done; } > $O/r4_fusion_bounds.log 2>&1
