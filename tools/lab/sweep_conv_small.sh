#!/bin/bash
# sweep conv blockings for the 16/32-channel layers: 2D and 3D
cd $GRAFT_REPO_ROOT
for cfg in "1 4" "1 2" "2 2" "2 4" "1 1" "2 1"; do
  set -- $cfg
  echo "== 2D NT=$1 MR=$2"
  CHAP_CONV_MINC=16 CHAP_CONV_NT=$1 CHAP_CONV_MR=$2 python tools/shape_table.py --config 2d --only "conv_fwd 2D k3 s1" 2>&1 | grep -E "conv_fwd +2D k3 s1 (16|32|16\+16)[^0-9+]" | cut -c1-125
done > gpurun_out/conv_sweep2d_small.log 2>&1
for cfg in "1 4" "2 4" "1 1" "2 1"; do
  set -- $cfg
  echo "== 3D NT=$1 MR=$2"
  CHAP_CONV_MINC=16 CHAP_CONV_NT=$1 CHAP_CONV_MR=$2 python tools/shape_table.py --config 3d --only "conv_fwd 3D k3 s1" 2>&1 | grep -E "conv_fwd +3D k3 s1 (16|32)[^0-9]" | cut -c1-125
done > gpurun_out/conv_sweep3d_small.log 2>&1
echo done
