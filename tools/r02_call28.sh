#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call28; mkdir -p "$out"
timeout 400 python -m pytest tests/test_widen_gpu.py tests/test_host_gpu.py -q --no-header -p no:cacheprovider -k "gru or update or glo" 2>&1 | grep -v "^$" | tail -12 | tee $out/tests.txt
GOSLAM_GRU_GLO_FUSED=0 timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab_unfused.json
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
