#!/bin/bash
# Final evidence of the round for the committed tree: kernel stats (2D / 3D) and the two bench lines.
R=$GRAFT_REPO_ROOT; cd $R
bash tools/lab/collect_r03.sh stats > gpurun_out/collect_stats.log 2>&1; tail -3 gpurun_out/collect_stats.log
python3 bench.py > gpurun_out/r3_bench_final4.json 2> gpurun_out/r3_bench_final4.err
python3 bench.py --config 3d --steps 20 --warmup 5 > gpurun_out/r3_bench3d_final4.json 2> gpurun_out/r3_bench3d_final4.err
python3 - <<'P'
import json
for f in ("gpurun_out/r3_bench_final4.json", "gpurun_out/r3_bench3d_final4.json"):
    d = json.load(open(f)); print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["top_kernel"]["frac"], [(e["dtype"], e["ms_per_step"]) for e in d.get("extra_configs", [])])
P
