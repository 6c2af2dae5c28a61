#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call38; mkdir -p "$out"
timeout 800 python -m pytest tests/test_widen_gpu.py tests/test_host_gpu.py tests/test_benchshape_gpu.py -q --no-header -p no:cacheprovider -x 2>&1 | grep -v "^$" | tail -8 | tee $out/tests.txt
timeout 60 python tools/glo_bench.py 2>&1 | tail -1 | tee $out/glo.json
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
