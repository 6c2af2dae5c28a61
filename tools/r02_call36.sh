#!/bin/bash
# functional smoke of the N > 1 bench path on one GPU (gloo, both ranks on cuda:0) + the tests touched since the last full run
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call36; mkdir -p "$out"
cd $R
GS_BENCH_SMOKE_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench2.log 2>&1
echo "rc=$?" | tee $out/rc.txt
tail -1 $out/bench2.log | cut -c1-1500
timeout 400 python -m pytest tests/test_neus_gpu.py tests/test_track_gpu.py -q --no-header -p no:cacheprovider -k "fused_mapper or ba_status or flat_adamw" 2>&1 | grep -v "^$" | tail -5 | tee $out/tests.txt
