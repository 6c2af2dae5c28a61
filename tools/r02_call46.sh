#!/bin/bash
# kernel trace of the bench command, summarised over the timed region
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call46; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python $R/bench.py --steps 10 --warmup 3 > $out/bench_traced.log 2>&1
f=$(ls /tmp/prof/*/*kernel_trace.csv /tmp/prof/*kernel_trace.csv 2>/dev/null | head -1)
cd $R
python tools/summarize_trace.py "$f" --steps 10 --warmup 3 > $out/tracking_kernel_stats.md 2>$out/summarize.err
head -40 $out/tracking_kernel_stats.md; tail -3 $out/summarize.err
tail -1 $out/bench_traced.log | cut -c1-200
