#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call33; mkdir -p "$out"
timeout 200 python tools/conv3x3_bench.py S480 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('persist', {k:(v['pp_ms'],v['pp_tflops']) for k,v in d.items() if isinstance(v,dict)})" | tee $out/bench_p.txt
GOSLAM_CONV3X3_PERSIST=0 timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab_np.json
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
