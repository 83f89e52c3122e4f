#!/bin/bash
# prefetch-depth sweep of the deep 2D conv instances: tools/lab/run_lab_pd.sh "1 2 3"
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/lab
for a in ${1:-1 2}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ichap_amd/csrc -DCHAP_CONV_PD=$a ${LABFLAGS} tools/lab/conv_lab.hip -w -o gpurun_out/lab/conv_lab_pd$a &
done
wait
for a in ${1:-1 2}; do echo "== PD=$a"; ./gpurun_out/lab/conv_lab_pd$a ${LABARGS:-d}; done
