#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call18; mkdir -p "$out"
timeout 300 python tools/conv3x3_pp_probe.py 2>&1 | tail -3 | tee $out/probe.json
