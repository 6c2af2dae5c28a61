#!/bin/bash
# Verifies and times every opt-in variant of the implicit-GEMM 3x3 convolution on an MI355X (one gpurun call, ~3 min):
#   plain / row-stacked tiling  x  lane permutation (bank-conflict-free B reads)  x  XCD-aware workgroup order  x
#   ConvGRU gate arithmetic fused into the convolutions' epilogues.
# Each configuration first runs the conv parity tests, then the layer micro-benchmark on the three map shapes, then the
# end-to-end keyframe A/B.  Results: gpurun_out/conv3x3_variants/*.json (copy what matters to profiles/).
#
#   /usr/local/graft/bin/gpurun --timeout 400 -- 'bash tools/conv3x3_variants.sh'
set -u
out=gpurun_out/conv3x3_variants
mkdir -p "$out"
run() {   # name, env assignments...
  local name=$1; shift
  echo "== $name ($*)"
  env "$@" timeout 90 python -m pytest tests/test_widen_gpu.py -q -x -k "conv3x3 or fused_gru or fused_bias" 2>&1 | tail -2 | tee "$out/$name.tests.txt"
  env "$@" timeout 120 python tools/conv3x3_bench.py all 2>/dev/null | tee "$out/$name.bench.json"
  env "$@" timeout 60 python tools/update_ab.py 2>/dev/null | tail -1 | tee "$out/$name.update_ab.json"
}
export GOSLAM_TEST_EXPERIMENTAL=1
run base GOSLAM_CONV3X3_LANEPERM=0
run laneperm GOSLAM_CONV3X3_LANEPERM=1
run laneperm_xcd GOSLAM_CONV3X3_LANEPERM=1 GOSLAM_CONV3X3_XCD=1
run gru_fused GOSLAM_GRU_FUSED=1
run gru_fused_laneperm_xcd GOSLAM_GRU_FUSED=1 GOSLAM_CONV3X3_LANEPERM=1 GOSLAM_CONV3X3_XCD=1
run stacked GOSLAM_CONV3X3_STACKED=1
run stacked_laneperm_xcd GOSLAM_CONV3X3_STACKED=1 GOSLAM_CONV3X3_LANEPERM=1 GOSLAM_CONV3X3_XCD=1
