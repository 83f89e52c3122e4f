#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
for nb in 512 1024 2048; do
  for c in 3d 2d; do
    CHAP_ACTBWD_BLOCKS=$nb python3 tools/shape_table.py --config $c --only "act_bwd C=16 @" --out $O/s_${c}_$nb.csv > $O/s_${c}_$nb.log 2>&1
    awk -F, -v nb=$nb 'NR>1{printf "blocks=%s %s %s us=%s GBps=%s | %s\n",nb,$1,$2,$4,$8,$13}' $O/s_${c}_$nb.csv
  done
done
