#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call11; mkdir -p "$out"
timeout 100 python tools/debug_version.py 2>&1 | tail -4 | tee $out/version.txt
