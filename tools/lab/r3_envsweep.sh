#!/bin/bash
# Round 3: runtime knobs of the HIP graph executor / queues on the whole iteration (default graph shape)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/env; mkdir -p $O; cd $R
b() { tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-extra $BARGS > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -3 $O/$tag.err; return 0; }
  python3 - $tag $O/$tag.json <<'P'
import json,sys
d=json.load(open(sys.argv[2])); print("%-40s %.3f ms  %.1f vol/s"%(sys.argv[1], d["ms_per_step"], d["value"]))
P
}
for cfg in 2d 3d; do
  if [ $cfg = 2d ]; then BARGS="--steps 30 --warmup 5"; else BARGS="--config 3d --steps 20 --warmup 5"; fi
  b ${cfg}_base CHAP_X=0
  b ${cfg}_hwq8 GPU_MAX_HW_QUEUES=8
  b ${cfg}_hwq2 GPU_MAX_HW_QUEUES=2
  b ${cfg}_pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  b ${cfg}_pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  b ${cfg}_batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
  b ${cfg}_batch16 DEBUG_HIP_GRAPH_BATCH_SIZE=16
  b ${cfg}_batch1024 DEBUG_HIP_GRAPH_BATCH_SIZE=1024
  b ${cfg}_devkernarg0 HIP_FORCE_DEV_KERNARG=0
  b ${cfg}_base2 CHAP_X=0
done
