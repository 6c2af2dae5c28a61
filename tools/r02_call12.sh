#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call12; mkdir -p "$out"
timeout 400 python -m pytest tests/test_neus_gpu.py tests/test_tcnn_dropin_gpu.py tests/test_widen_gpu.py -q --no-header -p no:cacheprovider -k "fused_mapper or fused_adamw or training or mapper_runs or contract or flat_adamw" 2>&1 | grep -v "^$" | tail -30 | cut -c1-300 | tee $out/tests.txt
