#!/bin/bash
# HBM and L2 request counters of the correlation-volume builder (8 edges x 60 x 80, tile8 + row-major)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call49; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum; do
  timeout 100 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o cv -- python $R/tools/corr_build_bench.py > $out/pmc_$c.log 2>&1 || echo "pass $c failed"
done
python - <<'PY' | tee $out/summary.txt
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r02_call49"
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "corr_volume" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v8 = [x for x in v]
        print(k, "launches", len(v), "min", min(v), "median", sorted(v)[len(v)//2], "max", max(v))
PY
