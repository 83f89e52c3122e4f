#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ct; mkdir -p $O; cd $R/tools/lab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHAP_CONV_TRACE -I$R/chap_amd/csrc -I$R/include -o /tmp/conv_lab_tr conv_lab.hip 2> $O/build.log || { tail -20 $O/build.log; exit 1; }
LAB_TRACE_ALL=1 timeout -k 10 120 /tmp/conv_lab_tr ${1:-d} > $O/trace_${1:-d}.log 2>&1
cut -c1-900 $O/trace_${1:-d}.log | tail -40
