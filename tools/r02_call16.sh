#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/r02_call16; mkdir -p "$out"
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 2>&1 | grep -v "^$" | tail -40 | cut -c1-250 | tee $out/tests.txt
timeout 300 python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o conv -- python $R/tools/profile_conv3x3.py > $out/pmc_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r02_call16"
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print(f"{k:30s} launches {n:3d}  per-launch {v/n:18.1f}")
PY
