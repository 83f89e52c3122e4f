#!/bin/bash
# round 4, GPU call 17: scratch fixes (conv_wp bias in LDS, source-pointer select) checked and timed; kernel-argument size +72 B per conv lane;
# upper bounds of folding bn_finalize / act_bwd_sum into their producers (lab skip knobs)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_bwd_gpu.py -x -q -k "conv or wgrad" > $O/c17_tests.log 2>&1 || { tail -5 $O/c17_tests.log; exit 1; }
ms() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
L=$O/r04_tail_bounds.log; : > $L
run() { # label, env...
  lab=$1; shift
  for c in 2d 3d; do
    st=30; [ $c = 3d ] && st=20
    echo "== $c $lab" >> $L
    env "$@" timeout -k 10 200 python3 bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | ms >> $L || exit 1
  done
}
for rep in 1 2; do
run "product" X=0 || exit 1
run "pad72 (conv argument block +72 B)" CHAP_LIBPATH=tools/lab/libchap_hip_pad72.so CHAP_CONV_PAD=72 || exit 1
run "lab" CHAP_LIBPATH=tools/lab/libchap_hip_lab.so || exit 1
run "lab skip bn_finalize" CHAP_LIBPATH=tools/lab/libchap_hip_lab.so CHAP_LAB_SKIP_BNFIN=1 || exit 1
run "lab skip act_bwd_sum" CHAP_LIBPATH=tools/lab/libchap_hip_lab.so CHAP_LAB_SKIP_ACTSUM=1 || exit 1
run "lab skip both" CHAP_LIBPATH=tools/lab/libchap_hip_lab.so CHAP_LAB_SKIP_BNFIN=1 CHAP_LAB_SKIP_ACTSUM=1 || exit 1
done
cd /tmp; export TMPDIR=/tmp
for k in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_dom2d_$k
  rocprofv3 --pmc $k --output-format csv -d $O/pmc_dom2d_$k -- python3 $R/tools/dominant_kernel.py 12 > $O/pmc_dom2d_$k.log 2>&1
done
(cd $R && python3 tools/pmc_traffic.py $O/pmc_dom2d_FETCH_SIZE $O/pmc_dom2d_WRITE_SIZE conv_ > $O/r04_pmc_traffic_dominant_2d.jsonl)
