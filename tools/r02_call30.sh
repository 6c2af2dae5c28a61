#!/bin/bash
# full GPU suite + bench line + kernel trace of the bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call30; mkdir -p "$out"
cd $R
timeout 900 python -m pytest tests -q --no-header -p no:cacheprovider -m gpu 2>&1 | grep -v "^$" | tail -15 | tee $out/tests.txt
timeout 400 python bench.py 2>$out/bench.err | tail -1 > $out/bench_line.json
cut -c1-400 $out/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python $R/bench.py --steps 10 --warmup 3 > $out/bench_traced.log 2>&1
f=$(ls /tmp/prof/*/*kernel_trace.csv /tmp/prof/*kernel_trace.csv 2>/dev/null | head -1)
cd $R
python tools/summarize_trace.py "$f" --steps 10 --warmup 3 > $out/tracking_kernel_stats.md 2>$out/summarize.err
head -30 $out/tracking_kernel_stats.md
