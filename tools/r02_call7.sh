#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call7
mkdir -p "$out"
timeout 400 python -m pytest tests/test_neus_gpu.py tests/test_widen_gpu.py tests/test_distributed_gpu.py -q --no-header -p no:cacheprovider -k "fused or flat_adamw or training or mapper or rccl" 2>&1 | tail -40 | tee $out/tests.txt
timeout 200 python tools/profile_neus_train.py 4096 > $out/neus_train_4096.txt 2>&1; tail -3 $out/neus_train_4096.txt
timeout 200 python tools/profile_neus_train.py 32768 > $out/neus_train_32768.txt 2>&1; tail -3 $out/neus_train_32768.txt
