#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call25; mkdir -p "$out"
timeout 400 python -m pytest tests/test_track_gpu.py tests/test_benchshape_gpu.py -q --no-header -p no:cacheprovider -k "corr or volume or pyramid" 2>&1 | grep -v "^$" | tail -8 | tee $out/tests.txt
timeout 200 python tools/corr_build_bench.py 2>&1 | tail -1 | tee $out/corr_build.json
