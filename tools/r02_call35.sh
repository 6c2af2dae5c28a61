#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call35; mkdir -p "$out"
timeout 500 python -m pytest tests/test_widen_gpu.py tests/test_benchshape_gpu.py -q --no-header -p no:cacheprovider -k "conv3x3 or fused or own_conv or update_operator" 2>&1 | grep -v "^$" | tail -12 | tee $out/tests.txt
timeout 200 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
timeout 100 python - <<'PY' 2>&1 | tail -2 | tee $out/flow2.json
import json, sys, torch
sys.path.insert(0, '.')
import bench
import go_slam_amd.droid_net as DN
dev = "cuda:0"
torch.backends.cudnn.benchmark = True
conv = torch.nn.Conv2d(128, 64, 3, padding=1).to(dev)
cache = DN._HalfWeights()
x = torch.randn(75, 128, 60, 80, device=dev).half().contiguous(memory_format=torch.channels_last)
hx = torch.empty((75, 320, 60, 80), device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
out = {}
for pp in (True, False, True, False):
    DN.CONV3X3_PP = pp
    out.setdefault("pp" if pp else "lib", []).append(round(1e3 * bench.time_op(lambda: DN.conv_bias_act(cache, conv, x, "relu", out=hx, out_channel=256), iters=30, warm=5), 1))
print(json.dumps(out))
PY
