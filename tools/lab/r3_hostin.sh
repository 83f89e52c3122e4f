#!/bin/bash
# PCIe-inclusive rate: every step is handed HOST tensors (resident = the headline mode; pinned / pageable = copied in front of the step;
# staged = ChapStep.stage(): the next batch travels on a copy stream beside the running iteration, what chap_amd's train() does)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hi; mkdir -p $O; cd $R
python3 -m pytest tests/test_train_step_gpu.py -x -q -k "staged or replay or train_entry" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python3 -m pytest tests/test_parity_gates_gpu.py -x -q -k "train_entry_point" > $O/tests2.log 2>&1 || { tail -20 $O/tests2.log; exit 1; }
tail -1 $O/tests2.log
for c in 2d 3d; do
  if [ $c = 2d ]; then A="--steps 50 --warmup 5"; else A="--config 3d --steps 20 --warmup 5"; fi
  for m in resident pinned pageable staged staged_pageable; do
    F=""; [ $m = pinned ] && F="--host-inputs pinned"; [ $m = pageable ] && F="--host-inputs pageable"; [ $m = staged ] && F="--host-inputs pinned --stage"; [ $m = staged_pageable ] && F="--host-inputs pageable --stage"
    python3 bench.py --no-cpu-baseline --no-extra $A $F > $O/${c}_$m.json 2> $O/${c}_$m.err || { tail -3 $O/${c}_$m.err; continue; }
    python3 -c "import json;d=json.load(open('$O/${c}_$m.json'));print('$c $m', d['ms_per_step'], d['value'])"
  done
done
