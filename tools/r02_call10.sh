#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call10; mkdir -p "$out"
timeout 200 python tools/debug_fused_mapper.py 2>&1 | tail -30 | tee $out/debug.txt
