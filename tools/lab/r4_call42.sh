#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
python3 tools/lab/ab_conv_bits.py save /tmp/a.pt 2>/dev/null || exit 1
CHAP_LIBPATH=tools/lab/libchap_hip_f32wbuf1.so python3 tools/lab/ab_conv_bits.py save /tmp/b.pt 2>/dev/null || exit 1
python3 tools/lab/ab_conv_bits.py cmp /tmp/a.pt /tmp/b.pt > $O/r04_wbuf1_bits.log 2>&1; cat $O/r04_wbuf1_bits.log
