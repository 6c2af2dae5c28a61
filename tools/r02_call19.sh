#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call19; mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I go_slam_amd/csrc -o /tmp/chol_bench tools/chol_bench.hip 2>&1 | tail -3
/tmp/chol_bench 144 | tee $out/chol144.txt
/tmp/chol_bench 150 | tee $out/chol150.txt
