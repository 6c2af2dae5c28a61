#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call42; mkdir -p "$out"
timeout 200 python tools/overlap_ab.py 2>&1 | tail -3 | tee $out/overlap_ab.json
timeout 400 python -m pytest tests/test_host_gpu.py tests/test_benchshape_gpu.py tests/test_widen_gpu.py -q --no-header -p no:cacheprovider -k "update or tracker or factor or fast_path or frontend" 2>&1 | grep -v "^$" | tail -5 | tee $out/tests.txt
