#!/bin/bash
# Round-2 first GPU call: conv3x3 variant A/B (tests + layer bench on 3 map shapes), end-to-end keyframe A/B for the
# opt-in switches, rocprofv3 kernel trace of the as-shipped bench, PMC passes on conv3x3_kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call1
mkdir -p "$out"
export GOSLAM_TEST_EXPERIMENTAL=1
tb() { local name=$1; shift
  echo "== $name ($*)"
  env "$@" timeout 120 python -m pytest tests/test_widen_gpu.py -q -x -k "conv3x3 or fused_gru or fused_bias" 2>&1 | tail -3 | tee "$out/$name.tests.txt"
  env "$@" timeout 150 python tools/conv3x3_bench.py all 2>/dev/null | tee "$out/$name.bench.json"
}
ab() { local name=$1; shift
  echo "== ab $name ($*)"
  env "$@" timeout 90 python tools/update_ab.py 2>/dev/null | tail -1 | tee "$out/$name.update_ab.json"
}
tb base GOSLAM_CONV3X3_LANEPERM=0
tb laneperm GOSLAM_CONV3X3_LANEPERM=1
tb laneperm_xcd GOSLAM_CONV3X3_LANEPERM=1 GOSLAM_CONV3X3_XCD=1
ab base GOSLAM_CONV3X3_LANEPERM=0
ab laneperm GOSLAM_CONV3X3_LANEPERM=1
ab gru_fused GOSLAM_GRU_FUSED=1
ab stacked GOSLAM_CONV3X3_STACKED=1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_traced.log 2>&1
tail -1 $out/bench_traced.log | head -c 3000
timeout 120 rocprofv3 --kernel-trace --stats -d $out/conv_trace -o conv -- python $R/tools/profile_conv3x3.py > $out/conv_trace.log 2>&1
pmc() { local name=$1; shift
  timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $out/conv_pmc_$name -o conv -- python $R/tools/profile_conv3x3.py > $out/conv_pmc_$name.log 2>&1 || echo "pmc pass $name failed: $*"
}
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
pmc sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
pmc sq4 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc sq5 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc grbm GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r02_call1"
for f in sorted(glob.glob(out + "/conv_pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print(f"{k:36s} launches {n:3d}  per-launch {v/n:18.1f}")
PY
# keep only the csv summaries (size cap on gpurun_out)
find $out -name "*.db" -delete 2>/dev/null
du -sh $out
ls -R $out | head -60
