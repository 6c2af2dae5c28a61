#!/bin/bash
# One round's evidence on an MI355X box (run through gpurun from the repository root):
#   bash tools/round_profile.sh r05
# full GPU suite, the bench line (driver flags), kernel traces of the tracking keyframe / the three mapping legs /
# MotionFilter.track / the global-BA stress step, the PMC passes of the mapper step, the Cholesky harness.
# Everything lands in gpurun_out/<tag>_final/; the summaries that are judged get copied to profiles/ by hand.
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${tag}_final
mkdir -p $OUT
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench.json
tail -c 1900 $OUT/bench.json; echo
# the N > 1 code path of bench.py on this 1-GPU box: 2 ranks on cuda:0, gloo carrying the collectives (functional smoke
# run of the sharded step + the per-rank exchange timing; never used for numbers)
GS_BENCH_SMOKE_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2> $OUT/bench_2ranks.err | tail -1 > $OUT/bench_2ranks_smoke.json
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_2ranks_smoke.json"))
    print("2-rank smoke:", d["n_gpus"], d["rccl_ranks"], d["collective_backend"], json.dumps(d["neus_train_weak"].get("exchange_per_rank"))[:600])
except Exception as exc:
    print("2-rank smoke FAILED:", repr(exc)); print(open("$OUT/bench_2ranks.err").read()[-1500:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_track -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_track.log 2>&1 || echo "prof track failed"
f=$(find $OUT/prof_track -name '*kernel_trace.csv' | head -1)
python $R/tools/summarize_trace.py $f --steps 5 --warmup 2 > $OUT/tracking_kernel_stats.md 2> $OUT/summarize.err
for leg in train train_weak render; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$leg -o t -- python $R/tools/profile_mapping.py $leg 10 > $OUT/prof_$leg.log 2>&1 || echo "prof $leg failed"
  f=$(find $OUT/prof_$leg -name '*kernel_trace.csv' | head -1)
  python $R/tools/summarize_kernels.py $f --steps 10 --title "profile_mapping.py $leg" > $OUT/mapping_${leg}_kernel_stats.md 2>> $OUT/summarize.err
done
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_mf -o t -- python $R/tools/profile_motion_filter.py 20 > $OUT/prof_mf.log 2>&1 || echo "prof mf failed"
f=$(find $OUT/prof_mf -name '*kernel_trace.csv' | head -1)
python $R/tools/summarize_kernels.py $f --steps 20 --after erfinv --title "MotionFilter.track, one 480x640 RGB-D input frame (20 steady-state frames; setup and warm frames excluded by the marker launch)" > $OUT/motion_filter_kernel_stats.md 2>> $OUT/summarize.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_e2e -o t -- python $R/tools/profile_frontend_e2e.py 16 > $OUT/prof_e2e.log 2>&1 || echo "prof e2e failed"
f=$(find $OUT/prof_e2e -name '*kernel_trace.csv' | head -1)
python $R/tools/summarize_kernels.py $f --steps 16 --after erfinv --title "end to end on a synthetic 640x480 RGB-D sequence, per keyframe: 4 x MotionFilter.track + Frontend.__call__ (window 25, max_factors 75; 16 steady-state keyframes after the marker launch)" > $OUT/frontend_e2e_kernel_stats.md 2>> $OUT/summarize.err
timeout 200 python $R/tools/profile_frontend_e2e.py 16 --ops > $OUT/frontend_e2e_ops.json 2> $OUT/frontend_e2e_ops.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_stress -o t -- python $R/tools/profile_stress.py 8 > $OUT/prof_stress.log 2>&1 || echo "prof stress failed"
f=$(find $OUT/prof_stress -name '*kernel_trace.csv' | head -1)
python $R/tools/summarize_kernels.py $f --steps 8 --after erfinv --title "global BA stress (200 keyframes, 1200 edges, 30x40): one update_lowmem(steps=8) invocation, per-edge caches cold, per step" > $OUT/stress_kernel_stats.md 2>> $OUT/summarize.err
find $OUT -name '*.csv' -delete
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" timeout 600 bash $R/tools/pmc_pass.sh $OUT/pmc_neus _kernel -- python $R/tools/profile_mapping.py train 3 > $OUT/pmc_neus.log 2>&1
# what the backward's pass 1 is bound by: instruction counts and busy cycles of the same launches
PMC_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS;SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES;SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQ_WAIT_INST_ANY SQ_WAIT_ANY" timeout 600 bash $R/tools/pmc_pass.sh $OUT/pmc_neus_sq _kernel -- python $R/tools/profile_mapping.py train 3 > $OUT/pmc_neus_sq.log 2>&1
[ -x $R/tools/chol_bench ] && timeout 150 $R/tools/chol_bench 150 192 198 294 300 306 342 360 450 456 1194 > $OUT/chol_bench.txt 2>&1
cd $R && timeout 900 bash $R/tools/gemm_comparator.sh $OUT/gemm > $OUT/gemm_comparator.log 2>&1
head -8 $OUT/tracking_kernel_stats.md; head -8 $OUT/stress_kernel_stats.md; head -4 $OUT/motion_filter_kernel_stats.md; head -4 $OUT/frontend_e2e_kernel_stats.md; tail -3 $OUT/summarize.err; tail -12 $OUT/chol_bench.txt
