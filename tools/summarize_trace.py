"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py` over the TIMED region only.

MIOpen's one-off solver search (naive_conv_* etc.) runs during warm-up and dominates the whole-process
--stats table, so this cuts the trace to the K timed keyframe steps: every FactorGraph.update issues
exactly one `ba_inputs_kernel`, preceded by one `reproject_kernel` at the start of the update.

usage: python tools/summarize_trace.py <kernel_trace.csv> --steps K --warmup W [--updates 6] > profiles/....md
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1)); s = name[len(m.group(0)):]
        return s[:n] + "<" + s[n:n + 24] + ">"
    m = re.match(r"_ZN2ck.*?(kernel_[a-z_0-9]+)", name)
    if m:
        return "ck::" + m.group(1)[:60]
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--warmup", type=int, required=True)
    ap.add_argument("--updates", type=int, default=6)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # one ba_inputs_kernel per FactorGraph.update (ba_prep_kernel runs once per edge set since the BA tables are cached)
    prep = [i for i, r in enumerate(rows) if "ba_inputs_kernel" in r["Kernel_Name"]]
    # bench.py order: [build_state] warm-up, TIMED, NeuS train bench, op breakdown -> the timed updates
    # are the last steps*updates BA calls before the first NeuS kernel.
    neus = next((i for i, r in enumerate(rows) if "neus_" in r["Kernel_Name"] or "render_sample" in r["Kernel_Name"]),
                len(rows))
    before = [k for k, i in enumerate(prep) if i < neus]
    last = len(before)
    first = last - a.steps * a.updates
    assert first >= a.warmup * a.updates, (first, len(prep))

    def update_start(k):     # index of the reproject_kernel that opens update k
        i = prep[k]
        while "reproject_kernel" not in rows[i]["Kernel_Name"]:
            i -= 1
        return i
    lo = update_start(first)
    hi = neus if neus < len(rows) else len(rows) - 1
    while hi > lo and "ba_update_kernel" not in rows[hi - 1]["Kernel_Name"] and "cvx_upsample" not in rows[hi - 1]["Kernel_Name"]:
        hi -= 1                      # end of the last timed update (host-side NeuS set-up follows)
    region = rows[lo:hi]
    wall = (int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in region:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        e = agg[short(r["Kernel_Name"])]
        e[0] += 1; e[1] += d
    busy = sum(v[1] for v in agg.values()) / 1e3
    print(f"timed region: {a.steps} keyframe steps x {a.updates} updates, {len(region)} dispatches, "
          f"wall {wall:.2f} ms ({wall / a.steps:.2f} ms/keyframe), GPU busy {busy:.2f} ms "
          f"({100 * busy / wall:.1f} % of wall)\n")
    print("| kernel | calls/keyframe | avg us | ms/keyframe | % of busy |")
    print("|---|---|---|---|---|")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"| `{k}` | {c / a.steps:.1f} | {t / c:.1f} | {t / 1e3 / a.steps:.3f} | {100 * t / 1e3 / busy:.1f} |")


if __name__ == "__main__":
    main()
