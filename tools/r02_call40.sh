#!/bin/bash
# clocks and power while the convolution runs back to back, and while the bench's keyframe loop runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call40; mkdir -p "$out"
rocm-smi --showmaxpower --showclocks --showpower 2>&1 | grep -v "^=\|^$" | head -20 | tee $out/idle.txt
sample() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' ' ; echo; sleep 0.25; done; }
echo "--- conv layers back to back" | tee $out/conv.txt
( sleep 6; sample 16 ) >> $out/conv.txt &
python tools/gru_layers_bench.py 2>&1 | tail -1 | tee -a $out/conv.txt
wait
echo "--- keyframe loop" | tee $out/kf.txt
( sleep 8; sample 16 ) >> $out/kf.txt &
python tools/update_ab.py 2>/dev/null | tail -1 | tee -a $out/kf.txt
wait
cat $out/conv.txt | cut -c1-220 | head -24
cat $out/kf.txt | cut -c1-220 | head -24
