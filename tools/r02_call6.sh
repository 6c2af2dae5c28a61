#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call6
mkdir -p "$out"
timeout 120 python tools/debug_fused_gru.py 2>&1 | tail -12 | tee $out/debug_fused.txt
timeout 200 python tools/conv3x3_pp_probe.py 2>&1 | tail -5 | tee $out/probe.json
timeout 200 python -m pytest tests/test_widen_gpu.py -q --no-header -p no:cacheprovider -k "pingpong or fused" 2>&1 | tail -5 | tee $out/tests.txt
timeout 120 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
