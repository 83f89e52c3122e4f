#!/bin/bash
# Round-2 evidence run on the GPU box: everything that ends up under profiles/ comes from this one script.
#   bash tools/lab/collect_profiles.sh        (from the repo root; writes gpurun_out/r02/*)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
# 1. bench lines: BASELINE config 1 (default), fp32 mode, 3D 112x112x80, 3D 112^3 extra row
python3 $R/bench.py > $O/bench2d_bf16.json 2> $O/bench2d_bf16.err
python3 $R/bench.py --dtype fp32 --no-cpu-baseline > $O/bench2d_fp32.json 2> $O/bench2d_fp32.err
python3 $R/bench.py --config 3d --steps 10 --warmup 3 > $O/bench3d_bf16.json 2> $O/bench3d_bf16.err
python3 $R/bench.py --config 3d --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench3d_fp32.json 2> $O/bench3d_fp32.err
python3 $R/bench.py --config 3d --size3d 112 112 112 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench3d_112cube_bf16.json 2> $O/bench3d_112cube.err
python3 $R/bench.py --config 3d --vat-iters 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench3d_k2_bf16.json 2> $O/bench3d_k2.err
CHAP_FORCE_DP=1 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench2d_dp1_fold.json 2> $O/bench2d_dp1_fold.err
CHAP_FORCE_DP=1 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --dp-overlap > $O/bench2d_dp1_overlap.json 2> $O/bench2d_dp1_overlap.err
echo "bench lines done"
# 2. kernel-trace stats of the two bench configurations
for c in 2d 3d; do
  rm -rf $O/ks$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$c -- python3 $R/bench.py --config $c --no-cpu-baseline --steps 10 --warmup 3 > $O/ks$c.json 2> $O/ks$c.err
  cp $(find $O/ks$c -name "*kernel_stats.csv" | head -1) $O/r02_bench${c}_kernel_stats.csv
done
echo "kernel stats done"
# 3. per-(kernel, layer shape) roofline table, with the instance names from a kernel trace of the same run
for c in 2d 3d; do
  rm -rf $O/st$c
  rocprofv3 --kernel-trace --output-format csv -d $O/st$c -- python3 $R/tools/shape_table.py --config $c --out $O/shapes$c.csv --trace-plan $O/plan$c.json > $O/shapes$c.log 2>&1
  (cd $R && python3 tools/shape_join.py $O/shapes$c.csv $O/plan$c.json $O/st$c $O/r02_conv_shapes_$c.csv)
done
python3 $R/tools/shape_table.py --config 2d --dtype fp32 --out $O/r02_conv_shapes_2d_fp32.csv > $O/shapes2d_fp32.log 2>&1
echo "shape tables done"
# 4. PMC passes (each alone: no trace flags): HBM traffic of the roofline kernels, MFMA busy of the conv / wgrad families
for c in 2d 3d; do
  a=""; [ $c = 3d ] && a="3d"
  rm -rf $O/pmc_fetch_$c $O/pmc_write_$c $O/pmc_mfma_$c
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$c -- python3 $R/tools/dominant_kernel.py $a > $O/pmc_fetch_$c.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$c -- python3 $R/tools/dominant_kernel.py $a > $O/pmc_write_$c.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_$c -- python3 $R/tools/shape_table.py --config $c --eager --reps 5 --only "k3 s1" > $O/pmc_mfma_$c.log 2>&1
  (cd $R && python3 tools/pmc_mfma.py $O/pmc_mfma_$c conv_fwd_kernel wgrad_kernel > $O/r02_mfma_util_$c.jsonl)
  (cd $R && python3 tools/pmc_summary.py $O/pmc_fetch_$c > $O/r02_pmc_fetch_$c.txt; python3 tools/pmc_summary.py $O/pmc_write_$c > $O/r02_pmc_write_$c.txt)
done
echo "pmc done"
# keep the merge small
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
ls $O | head -80
