#!/bin/bash
# rocprofv3 --pmc passes of one command, ONE counter group per pass (no trace domains alongside: MI355X_MICROARCH.md),
# then a per-kernel table of the counter medians.   usage: tools/pmc_pass.sh <out_dir> <kernel-substring> -- <command...>
set -u
out=$(realpath -m "$1"); pat=$2; shift 3        # (absolute: the passes run from /tmp; give the command absolute paths too)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# PMC_GROUPS="A B;C;D E" overrides the counter groups (';' between passes)
IFS=';' read -r -a groups <<< "${PMC_GROUPS:-FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum;TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum;TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum}"
for c in "${groups[@]}"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 180 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$tag -o p -- "$@" > $out/pmc_$tag.log 2>&1 || echo "pass $tag failed (see $out/pmc_$tag.log)"
done
python3 - "$out" "$pat" <<'PY'
import csv, glob, collections, json, sys
out, pat = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            v = sorted(v)
            res[k][c] = {"launches": len(v), "median": v[len(v) // 2], "min": v[0], "max": v[-1]}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
find $out -name '*.csv' -size +2M -delete
