#!/bin/bash
# Round-4 evidence run on the GPU box: what ends up under profiles/r04_* comes from this script.
#   bash tools/lab/collect_r04.sh [part ...]     parts: bench stats shapes pmc   (default: all)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PARTS=${*:-bench stats shapes pmc timeline}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
python3 $R/bench.py > $O/r04_bench2d_bf16.json 2> $O/bench2d_bf16.err
python3 $R/bench.py --config 3d --steps 20 --warmup 5 > $O/r04_bench3d_bf16.json 2> $O/bench3d_bf16.err
python3 $R/bench.py --dtype fp32 --no-cpu-baseline --no-extra > $O/r04_bench2d_fp32.json 2> $O/bench2d_fp32.err
python3 $R/bench.py --config 3d --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline > $O/r04_bench3d_fp32.json 2> $O/bench3d_fp32.err
python3 $R/bench.py --config 3d --size3d 112 112 112 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_bench3d_112cube_bf16.json 2> $O/bench3d_112cube.err
python3 $R/bench.py --config 3d --vat-iters 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_bench3d_k2_bf16.json 2> $O/bench3d_k2.err
CHAP_WGRAD_WP=0 CHAP_SPLIT_CONCAT=0 python3 $R/bench.py --no-cpu-baseline --no-extra > $O/r04_bench2d_bf16_r3kernels.json 2> $O/bench2d_r3kernels.err
echo "bench lines done"
fi
if has stats; then
for c in 2d 3d; do
  rm -rf $O/ks$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$c -- python3 $R/bench.py --config $c --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $O/ks$c.json 2> $O/ks$c.err
  cp $(find $O/ks$c -name "*kernel_stats.csv" | head -1) $O/r04_bench${c}_kernel_stats.csv
done
echo "kernel stats done"
fi
if has shapes; then
for c in 2d 3d; do
  rm -rf $O/st$c
  rocprofv3 --kernel-trace --output-format csv -d $O/st$c -- python3 $R/tools/shape_table.py --config $c --out $O/shapes$c.csv --trace-plan $O/plan$c.json > $O/shapes$c.log 2>&1
  (cd $R && python3 tools/shape_join.py $O/shapes$c.csv $O/plan$c.json $O/st$c $O/r04_conv_shapes_$c.csv)
done
echo "shape tables done"
fi
if has pmc; then
# HBM-side traffic (FETCH_SIZE and WRITE_SIZE in separate passes, no trace flags) of the kernels that own the time
pmc() { # tag, config, --only substring
  for k in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${1}_$k
    rocprofv3 --pmc $k --output-format csv -d $O/pmc_${1}_$k -- python3 $R/tools/shape_table.py --config $2 --eager --reps 5 --only "$3" > $O/pmc_${1}_$k.log 2>&1
  done
  (cd $R && python3 tools/pmc_traffic.py $O/pmc_${1}_FETCH_SIZE $O/pmc_${1}_WRITE_SIZE > $O/r04_pmc_traffic_$1.jsonl)
}
pmc wgrad16_2d 2d "wgrad 2D k3 s1 A=16 B=16 @256x256 N=12 A=ar"
pmc actbwd16_2d 2d "act_bwd C=16 @12x1x256x256 ng=1 bn=1 ar"
pmc wgrad1616_2d 2d "wgrad 2D k3 s1 A=16+16 B=16 @256x256 N=12"
pmc conv128_2d 2d "conv_fwd 2D k3 s1 128->128 @32x32 N=12 stats src=ar"
pmc conv64_3d 3d "conv_fwd 3D k3 s1 64->64 @28x28x20 N=2 stats src=ar"
for c in 2d 3d; do
  a=""; [ $c = 3d ] && a="3d"
  for k in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_dom${c}_$k
    rocprofv3 --pmc $k --output-format csv -d $O/pmc_dom${c}_$k -- python3 $R/tools/dominant_kernel.py $a > $O/pmc_dom${c}_$k.log 2>&1
  done
  (cd $R && python3 tools/pmc_traffic.py $O/pmc_dom${c}_FETCH_SIZE $O/pmc_dom${c}_WRITE_SIZE conv_ > $O/r04_pmc_traffic_dominant_$c.jsonl)
done
echo "pmc done"
fi
if has timeline; then
for c in 2d 3d; do
  (cd $R && CHAP_LIBPATH=tools/lab/libchap_hip_lab.so python3 tools/timeline_untraced.py --config $c --out $O/r04_timeline_untraced_$c.json > $O/timeline_$c.log 2>&1)
  (cd $R && CHAP_LIBPATH=tools/lab/libchap_hip_lab.so python3 tools/timeline_untraced.py --config $c --steady 1 --out $O/r04_timeline_untraced_${c}_isolated.json > $O/timeline_${c}_isolated.log 2>&1)
done
echo "timelines done"
fi
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +4M -delete
ls $O | head -100
