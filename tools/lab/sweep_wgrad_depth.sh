#!/bin/bash
# on the GPU box: rebuild the bf16 weight-gradient TU with different LDS pipeline depths and time the layer shapes
cd "$(dirname "$0")/../.."
R=$PWD; B=$R/chap_amd/csrc/_build
for d in ${1:-4 6 10}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -DCHAP_WGRAD_DEPTH=$d -c $R/chap_amd/csrc/wgrad_bf16.hip -o $B/wgrad_bf16.hip.o -w || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/chap_amd/libchap_hip.so $B/*.o || exit 1
  echo "== DEPTH $d"
  python tools/shape_table.py --config 3d --only wgrad 2>&1 | grep "k3 s1"
  python tools/shape_table.py --config 2d --only wgrad 2>&1 | grep "k3 s1"
done
