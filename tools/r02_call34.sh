#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call34; mkdir -p "$out"
timeout 600 python -m pytest tests/test_track_gpu.py tests/test_benchshape_gpu.py -q --no-header -p no:cacheprovider -k "ba" 2>&1 | grep -v "^$" | tail -8 | tee $out/tests.txt
timeout 200 python - <<'PY' 2>&1 | tail -3 | tee $out/ba.json
import json, torch, sys
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
video, op, graph, _ = bench.build_state(dev, seed=43)
br = bench.op_breakdown(video, op, graph)
print(json.dumps({k: v for k, v in br.items() if "ms" in k}))
PY
