#!/bin/bash
# four-wave (512-register) variant of the ping-pong convolution: parity + per-layer timing + keyframe A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call39; mkdir -p "$out"
export GOSLAM_CONV3X3_V3=1
timeout 500 python -m pytest tests/test_widen_gpu.py tests/test_benchshape_gpu.py -q --no-header -p no:cacheprovider -k "conv3x3 or fused or own_conv or update_operator" 2>&1 | grep -v "^$" | tail -12 | tee $out/tests.txt
timeout 200 python tools/conv3x3_bench.py S480 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('v3', {k:(v['pp_ms'],v['pp_tflops'],v['pp_max_abs_diff']) for k,v in d.items() if isinstance(v,dict)})" | tee $out/bench_v3.txt
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab_v3.json
export GOSLAM_CONV3X3_V3=0
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab_pp.json
