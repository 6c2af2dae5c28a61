#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call31; mkdir -p "$out"
timeout 500 python -m pytest tests/test_widen_gpu.py tests/test_host_gpu.py tests/test_track_gpu.py -q --no-header -p no:cacheprovider -k "gru or update or glo or conv1x1 or ba_status" 2>&1 | grep -v "^$" | tail -12 | tee $out/tests.txt
timeout 100 python tools/conv1x1_bench.py 2>&1 | tail -1 | tee $out/conv1x1.json
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
