#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call17; mkdir -p "$out"
timeout 200 python tools/debug_bench_gap.py 2>&1 | tail -8 | tee $out/gap.txt
timeout 200 python -m pytest tests/test_widen_gpu.py -q --no-header -p no:cacheprovider -k "edge_proposal" 2>&1 | grep -v "^$" | tail -12 | cut -c1-250 | tee $out/tests.txt
