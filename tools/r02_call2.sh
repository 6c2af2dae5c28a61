#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call2
mkdir -p "$out"
timeout 600 python -m pytest tests/test_tcnn_dropin_gpu.py tests/test_benchshape_gpu.py -q -x --no-header -p no:cacheprovider 2>&1 | tail -60 | tee $out/tests.txt
GOSLAM_TEST_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_widen_gpu.py -q -k "fused_gru" --no-header 2>&1 | tail -40 | tee $out/fused_gru.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_traced.log 2>&1
ls -la $out/trace | head
