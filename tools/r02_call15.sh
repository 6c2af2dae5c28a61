#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call15; mkdir -p "$out"
timeout 300 python -m pytest tests/test_widen_gpu.py tests/test_neus_gpu.py -q --no-header -p no:cacheprovider -k "pingpong or fused" 2>&1 | grep -v "^$" | tail -8 | cut -c1-300 | tee $out/tests.txt
timeout 200 python tools/conv3x3_pp_probe.py 2>&1 | tail -3 | tee $out/probe.json
timeout 120 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
