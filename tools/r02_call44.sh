#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call44; mkdir -p "$out"
timeout 600 python -m pytest tests/test_widen_gpu.py tests/test_benchshape_gpu.py tests/test_host_gpu.py -q --no-header -p no:cacheprovider -k "gru or fused or update or tracker" 2>&1 | grep -v "^$" | tail -6 | tee $out/tests.txt
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
