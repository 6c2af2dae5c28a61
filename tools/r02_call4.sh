#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call4
mkdir -p "$out"
timeout 300 python -m pytest tests/test_widen_gpu.py tests/test_multiprocess_gpu.py -q --no-header -p no:cacheprovider -k "pingpong or two_processes" 2>&1 | tail -40 | tee $out/tests.txt
timeout 200 python tools/conv3x3_bench.py all 2>/dev/null | tee $out/conv_bench.json
timeout 120 python tools/update_ab.py 2>/dev/null | tail -1 | tee $out/update_ab.json
cd /tmp && export TMPDIR=/tmp
pmc() { local name=$1; shift
  GOSLAM_CONV3X3_PP=1 timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o conv -- python $R/tools/profile_conv3x3.py > $out/pmc_$name.log 2>&1 || echo "pmc pass $name failed: $*"
}
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
pmc sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
pmc grbm GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r02_call4"
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"] + " " + r["Kernel_Name"][:60]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print(f"{k:100s} launches {n:3d}  per-launch {v/n:18.1f}")
PY
