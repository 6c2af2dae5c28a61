#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call9
mkdir -p "$out"
timeout 400 python -m pytest tests/test_neus_gpu.py tests/test_widen_gpu.py tests/test_host_gpu.py -q --no-header -p no:cacheprovider -k "fused or update_module or update_operator or gru" 2>&1 | grep -v "^$" | tail -40 | cut -c1-300 | tee $out/tests.txt
timeout 120 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
