#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call23; mkdir -p "$out"
timeout 300 python -m pytest tests/test_widen_gpu.py -q --no-header -p no:cacheprovider -k "conv7x7" 2>&1 | grep -v "^$" | tail -8 | tee $out/tests.txt
timeout 200 python tools/conv7x7_bench.py 2>&1 | tail -1 | tee $out/conv7x7.json
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
