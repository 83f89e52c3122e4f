#!/bin/bash
# builds and runs the conv ablation lab on the GPU box: tools/lab/run_lab.sh "0 1 2 4 8 16"
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/lab
for a in ${1:-0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ichap_amd/csrc -DCHAP_ABLATE=$a ${LABFLAGS} tools/lab/conv_lab.hip -w -o gpurun_out/lab/conv_lab_$a &
done
wait
for a in ${1:-0}; do ./gpurun_out/lab/conv_lab_$a ${LABARGS}; done
