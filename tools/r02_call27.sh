#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call27; mkdir -p "$out"
for d in 0 1 2 3; do echo "dbg $d"; GS_CORR_DBG=$d timeout 200 python tools/corr_build_bench.py 2>&1 | tail -1 | cut -c1-330; done | tee $out/dbg.txt
