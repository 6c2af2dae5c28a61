#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call32; mkdir -p "$out"
timeout 500 python -m pytest tests/test_widen_gpu.py -q --no-header -p no:cacheprovider -k "conv3x3 or fused or own_conv" 2>&1 | grep -v "^$" | tail -12 | tee $out/tests.txt
GOSLAM_CONV3X3_PERSIST=0 timeout 200 python tools/conv3x3_bench.py S480 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('nonpersist', {k:(v['pp_ms'],v['pp_tflops']) for k,v in d.items() if isinstance(v,dict)})" | tee $out/bench_np.txt
timeout 200 python tools/conv3x3_bench.py S480 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('persist', {k:(v['pp_ms'],v['pp_tflops']) for k,v in d.items() if isinstance(v,dict)})" | tee $out/bench_p.txt
GOSLAM_CONV3X3_PERSIST=0 timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab_np.json
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
