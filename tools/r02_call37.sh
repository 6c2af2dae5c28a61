#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call37; mkdir -p "$out"
timeout 500 python -m pytest tests/test_widen_gpu.py tests/test_host_gpu.py -q --no-header -p no:cacheprovider -k "glo or gru or update_module" 2>&1 | grep -v "^$" | tail -5 | tee $out/tests.txt
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
