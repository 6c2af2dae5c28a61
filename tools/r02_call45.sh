#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call45; mkdir -p "$out"
timeout 600 python -m pytest tests/test_track_gpu.py tests/test_benchshape_gpu.py tests/test_host_gpu.py -q --no-header -p no:cacheprovider -k "ba or update or tracker or damping or factor" 2>&1 | grep -v "^$" | tail -6 | tee $out/tests.txt
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
