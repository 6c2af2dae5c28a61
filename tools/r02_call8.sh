#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call8
mkdir -p "$out"
timeout 120 python tools/debug_fused_gru.py 2>&1 | tail -40 | tee $out/debug_fused.txt
timeout 300 python -m pytest tests/test_neus_gpu.py tests/test_widen_gpu.py -q --no-header -p no:cacheprovider -k "fused_mapper or build_rays or encoders_on_device" 2>&1 | grep -v "^$" | tail -60 | cut -c1-250 | tee $out/tests.txt
