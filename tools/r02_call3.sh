#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call3
mkdir -p "$out"
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=15 2>&1 | tail -150 | tee $out/tests.txt
