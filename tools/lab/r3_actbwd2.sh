#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab2; mkdir -p $O; cd $R
python3 -m pytest tests/test_kernels_bwd_gpu.py tests/test_kernels_gpu.py tests/test_net2d_gpu.py tests/test_net3d_gpu.py tests/test_train_step_gpu.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for c in 3d 2d; do python3 tools/shape_table.py --config $c --only "act_bwd C=" --out $O/s_$c.csv > $O/s_$c.log 2>&1; awk -F, 'NR>1{printf "%s %s us=%s GBps=%s | %s\n",$1,$2,$4,$8,$13}' $O/s_$c.csv; done
for rep in 1 2; do
python3 bench.py --no-cpu-baseline --no-extra --config 3d --steps 20 --warmup 5 > $O/b3d.json 2>$O/b3d.err; python3 -c "import json;d=json.load(open('$O/b3d.json'));print('3d',d['ms_per_step'],d['value'])"
python3 bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 > $O/b2d.json 2>$O/b2d.err; python3 -c "import json;d=json.load(open('$O/b2d.json'));print('2d',d['ms_per_step'],d['value'])"
done
