#!/bin/bash
# round 4, GPU call 46: smoke(), a short bench line of each configuration, and the PMC passes of the final tree (3D 64->64 conv: what the register bound's scratch costs in HBM bytes; dominant kernels)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/lab/collect_r04.sh pmc > $O/collect_pmc.log 2>&1; tail -2 $O/collect_pmc.log
cat $O/r04_pmc_traffic_conv64_3d.jsonl | cut -c1-300; cat $O/r04_pmc_traffic_dominant_3d.jsonl | cut -c1-300; cat $O/r04_pmc_traffic_dominant_2d.jsonl | cut -c1-300
