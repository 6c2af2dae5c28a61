#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call47; mkdir -p "$out"
timeout 200 python tools/conv3x3_pp_probe.py 2>/dev/null | tail -3 > $out/probe.jsonl
cut -c1-200 $out/probe.jsonl
timeout 500 python -m pytest tests/test_widen_gpu.py tests/test_benchshape_gpu.py -q --no-header -p no:cacheprovider -k "conv3x3 or fused or own_conv or update_operator" 2>&1 | grep -v "^$" | tail -4 | tee $out/tests.txt
