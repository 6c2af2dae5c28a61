#!/bin/bash
# final state: full GPU suite + smoke + default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call48; mkdir -p "$out"
cd $R
timeout 900 python -m pytest tests -q --no-header -p no:cacheprovider -m gpu 2>&1 | grep -v "^$" | tail -4 | tee $out/tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
timeout 300 python bench.py --no-cpu-baseline 2>$out/bench.err | tail -1 > $out/bench_line.json
cut -c1-330 $out/bench_line.json
