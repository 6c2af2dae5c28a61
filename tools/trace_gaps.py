"""Where the GPU idles: gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV, attributed to the pair (kernel
before, kernel after) and summed per step:
    python tools/trace_gaps.py <kernel_trace.csv> --steps N [--after MARKER] [--min-us 20]"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)(?:ILi|ILb|E[A-Z]|I[A-Z])", name)
    if m:
        name = m.group(1)
    return re.split(r"[<(]", name)[0][:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--after", default="")
    ap.add_argument("--min-us", type=float, default=20.0)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
    if a.after:
        marks = [i for i, r in enumerate(rows) if a.after in r["Kernel_Name"]]
        if marks:
            rows = rows[marks[-1] + 1:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    busy = idle_all = idle_big = 0.0
    end = None
    prev = ""
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += (e - s) / 1e3
        if end is not None and s > end:
            g = (s - end) / 1e3
            idle_all += g
            if g >= a.min_us:
                idle_big += g
                k = agg[(prev, short(r["Kernel_Name"]))]
                k[0] += 1
                k[1] += g
        end = e if end is None else max(end, e)
        prev = short(r["Kernel_Name"])
    n = a.steps
    print(f"{len(rows)} dispatches / {n} steps: busy {busy / n / 1e3:.3f} ms, idle {idle_all / n / 1e3:.3f} ms per step "
          f"({idle_big / n / 1e3:.3f} ms of it in gaps >= {a.min_us:g} us)\n")
    print("| kernel before the gap | kernel after | gaps/step | avg us | us/step |")
    print("|---|---|---|---|---|")
    for (p, q), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"| `{p}` | `{q}` | {c / n:.2f} | {t / c:.0f} | {t / n:.0f} |")


if __name__ == "__main__":
    main()
