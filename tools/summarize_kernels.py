"""Per-kernel table from a rocprofv3 --kernel-trace CSV of a command that repeats the same step `--steps` times:
    python tools/summarize_kernels.py <kernel_trace.csv> --steps N [--title T] > profiles/....md
(also --pmc counter_collection CSVs with --counter NAME: sums the counter per kernel per step)."""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    return name[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--title", default="")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--after", default="", help="count only the dispatches after the LAST kernel whose name contains this "
                                                "substring (a marker launch that separates one-off setup from the steps)")
    a = ap.parse_args()
    rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
    if a.after:
        marks = [i for i, r in enumerate(rows) if a.after in r["Kernel_Name"]]
        if marks:
            rows = rows[marks[-1] + 1:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        e = agg[short(r["Kernel_Name"])]
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy = sum(v[1] for v in agg.values())
    print(f"{a.title}: {len(rows)} dispatches over {a.steps} identical steps = {len(rows) / a.steps:.1f} launches/step, "
          f"GPU busy {busy / a.steps / 1e3:.3f} ms/step\n")
    print("| kernel | calls/step | avg us | us/step | % of busy |")
    print("|---|---|---|---|---|")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"| `{k}` | {c / a.steps:.1f} | {t / c:.1f} | {t / a.steps:.1f} | {100 * t / busy:.1f} |")


if __name__ == "__main__":
    main()
