#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call26; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o cb -- python $R/tools/corr_build_bench.py > $out/log.txt 2>&1
f=$(ls /tmp/prof/*kernel_stats.csv /tmp/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
head -12 "$f" | cut -c1-200 | tee $out/kernel_stats.txt
