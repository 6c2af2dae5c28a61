#!/bin/bash
# SQ counters of the two schedules of the ping-pong convolution on ONE box (z|r layer, plain epilogue) + kernel times
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r02_call41
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pmc() { local v3=$1 name=$2; shift; shift
  GOSLAM_CONV3X3_V3=$v3 timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_v${v3}_$name -o conv -- python $R/tools/profile_conv3x3.py > $out/pmc_v${v3}_$name.log 2>&1 || echo "pmc pass $name failed: $*"
}
for v in 0 1; do
  pmc $v sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
  pmc $v sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
  pmc $v sq3 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
  pmc $v grbm GRBM_GUI_ACTIVE
  GOSLAM_CONV3X3_V3=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_v$v -o conv -- python $R/tools/profile_conv3x3.py > $out/trace_v$v.log 2>&1
done
python - <<'PY' | tee $out/summary.txt
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r02_call41"
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"] + " " + r["Kernel_Name"][17:45]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print(f"{k:70s} launches {n:3d}  per-launch {v/n:18.1f}")
for f in sorted(glob.glob(out + "/trace_*/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Name"]:
            print(f.split("/")[-2][:10], r["Name"][17:45], "calls", r["Calls"], "avg_ns", r["AverageNs"])
PY
