#!/bin/bash
# Wave-cycle breakdown (tools/pmc_wave_breakdown.py) of the kernels that own the iteration's time; one --pmc pass each, no trace flags.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CTR="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
one() { # tag config only-substring
  rm -rf $O/$1
  rocprofv3 --pmc $CTR --output-format csv -d $O/$1 -- python3 $R/tools/shape_table.py --config $2 --eager --reps 5 --only "$3" > $O/$1.log 2>&1 || { tail -5 $O/$1.log; return 0; }
  (cd $R && python3 tools/pmc_wave_breakdown.py $O/$1 --last 5 conv_fwd conv_kpar wgrad_kernel act_bwd_kernel upsample | sed "s/^/{\"shape\": \"$3\", \"r\": /; s/$/}/") >> $O/wave_breakdown.jsonl
  find $O/$1 -name "*counter_collection.csv" -size +4M -delete
  echo "$1 done"
}
rm -f $O/wave_breakdown.jsonl
one c128_2d 2d "conv_fwd 2D k3 s1 128->128 @32x32 N=12 stats src=ar"
one c16_2d 2d "conv_fwd 2D k3 s1 16->16 @256x256 N=12 stats src=ar"
one w16_2d 2d "wgrad 2D k3 s1 A=16 B=16 @256x256 N=12 A=ar"
one a16_2d 2d "act_bwd C=16 @12x1x256x256 ng=1 bn=1 ar"
one c64_3d 3d "conv_fwd 3D k3 s1 64->64 @28x28x20 N=2 stats src=ar"
one c16_3d 3d "conv_fwd 3D k3 s1 16->16 @112x112x80 N=2 stats src=-"
one k128_3d 3d "conv_fwd 3D k3 s1 128->128 @14x14x10 N=2 stats src=ar"
cat $O/wave_breakdown.jsonl
