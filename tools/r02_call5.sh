#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call5
mkdir -p "$out"
timeout 300 python -m pytest tests/test_widen_gpu.py tests/test_host_gpu.py tests/test_benchshape_gpu.py -q --no-header -p no:cacheprovider -k "pingpong or fused or conv3x3 or update_module or update_operator" 2>&1 | tail -40 | tee $out/tests.txt
timeout 120 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
GOSLAM_GRU_FUSED=0 timeout 120 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab_unfused.json
timeout 200 python tools/profile_neus_train.py 4096 > $out/neus_train_4096.txt 2>&1; tail -5 $out/neus_train_4096.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_traced.log 2>&1
grep -o '{"metric".*' $out/bench_traced.log | head -c 1500
rm -f $out/trace/*domain_stats* ; ls -la $out/trace | head
