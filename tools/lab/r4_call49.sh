#!/bin/bash
# round 4, GPU call 49: the Dice gate with its re-measured threshold, and the two bench lines of the round's last commit
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_parity_gates_gpu.py -x -q -k "dice_gate" 2>&1 | tail -2
python3 bench.py > $O/r04_bench2d_bf16.json 2> $O/bench2d_bf16.err; cut -c1-330 $O/r04_bench2d_bf16.json; echo
python3 bench.py --config 3d --steps 20 --warmup 5 > $O/r04_bench3d_bf16.json 2> $O/bench3d_bf16.err; cut -c1-330 $O/r04_bench3d_bf16.json
