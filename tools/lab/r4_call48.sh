#!/bin/bash
# round 4, GPU call 48: the paired bf16 / fp32 Dice statistic once more, on the round's last tree (the bf16 kernels changed after the first run)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/r04; cd $R
timeout -k 10 1000 python3 tools/dice_pairs.py --seeds 6 --out $O/r04_dice_pairs.json > $O/r04/dice_pairs.log 2>&1; rc=$?
tail -12 $O/r04/dice_pairs.log | cut -c1-300; exit $rc
