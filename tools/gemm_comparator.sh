#!/bin/bash
# conv3x3_pp against a plain library GEMM of its own implicit-GEMM shape, SAME box, SAME counters:
#   bash tools/gemm_comparator.sh gpurun_out/r06_gemm     -> <out>/summary.json
# per kernel: event-timed median, SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE (-> effective clock = cycles / wall) in separate
# rocprofv3 --pmc passes (no trace domains alongside), one --kernel-trace pass for the durations.
out=$(realpath -m "${1:-gpurun_out/r06_gemm}"); R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $out
for g in gru_zr gru_q square; do python $R/tools/profile_gemm_comparator.py $g 8 > $out/gemm_$g.json 2> $out/gemm_$g.err; tail -1 $out/gemm_$g.json; done
G="SQ_VALU_MFMA_BUSY_CYCLES;GRBM_GUI_ACTIVE;SQ_BUSY_CYCLES SQ_WAVE_CYCLES;SQ_WAIT_INST_ANY SQ_INSTS_VALU"
PMC_GROUPS="$G" timeout 600 bash $R/tools/pmc_pass.sh $out/pmc_conv conv3x3_pp -- python $R/tools/profile_conv3x3.py gru_zr > $out/pmc_conv.log 2>&1
PMC_GROUPS="$G" timeout 600 bash $R/tools/pmc_pass.sh $out/pmc_gemm_zr "" -- python $R/tools/profile_gemm_comparator.py gru_zr 8 > $out/pmc_gemm_zr.log 2>&1
PMC_GROUPS="$G" timeout 600 bash $R/tools/pmc_pass.sh $out/pmc_gemm_sq "" -- python $R/tools/profile_gemm_comparator.py square 8 > $out/pmc_gemm_sq.log 2>&1
cd /tmp && export TMPDIR=/tmp
for t in "conv python $R/tools/profile_conv3x3.py gru_zr" "gemm_zr python $R/tools/profile_gemm_comparator.py gru_zr 8" "gemm_sq python $R/tools/profile_gemm_comparator.py square 8"; do
  set -- $t; tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$tag -o t -- "$@" > $out/trace_$tag.log 2>&1
  f=$(find $out/trace_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -6 $f > $out/trace_$tag.stats.csv
done
find $out -name '*.csv' -size +1M -delete
python3 - $out <<'PY'
import json, sys, glob, csv, os
out = sys.argv[1]
res = {}
for tag in ("conv", "gemm_zr", "gemm_sq"):
    p = f"{out}/pmc_{tag}/pmc_summary.json"
    pm = json.load(open(p)) if os.path.exists(p) else {}
    # the dominant kernel = the one with the most MFMA-busy cycles
    best = max(pm.items(), key=lambda kv: kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("median", 0), default=(None, {}))
    st = f"{out}/trace_{tag}.stats.csv"
    dur = None
    if os.path.exists(st):
        rows = list(csv.DictReader(open(st)))
        if rows:
            r0 = max(rows, key=lambda r: float(r.get("TotalDurationNs", 0) or 0))
            dur = {"kernel": r0.get("Name", "")[:90], "avg_us": float(r0.get("AverageNs", 0)) / 1e3, "calls": int(r0.get("Calls", 0))}
    res[tag] = {"pmc_kernel": best[0], "counters": {k: v["median"] for k, v in best[1].items()}, "trace": dur}
    c = res[tag]["counters"]
    if dur and "GRBM_GUI_ACTIVE" in c:
        us = dur["avg_us"]
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x 4 (quad-cycle units as in r05)
        res[tag]["derived"] = {"cycles_per_xcd": c["GRBM_GUI_ACTIVE"] / 8.0, "effective_clock_GHz": c["GRBM_GUI_ACTIVE"] / 8.0 / us / 1e3,
                               "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)}
for g in ("gru_zr", "gru_q", "square"):
    try:
        res["event_" + g] = json.loads(open(f"{out}/gemm_{g}.json").read().strip().splitlines()[-1])
    except Exception as exc:
        res["event_" + g] = {"error": repr(exc)}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:4000])
PY
