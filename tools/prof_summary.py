"""Summarise rocprofv3 CSV output (kernel trace and/or PMC counter collection) per kernel name.

    python tools/prof_summary.py <dir> [<out.txt>]
Reads every *kernel_trace.csv / *counter_collection.csv under <dir>, prints per-kernel launch count,
total/avg duration and per-launch mean of each counter (sum over dimensions).  Small enough to commit
under profiles/.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    name = re.sub(r"\(.*$", "", name)
    m = re.search(r"md_gemm_conv_kernel<GCfg<([^>]*)>", name)
    if m:
        return "md_gemm_conv_kernel<" + m.group(1).replace(" ", "") + ">"
    return name[:110]


def main():
    root = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    files = glob.glob(os.path.join(root, "**", "*.csv"), recursive=True)
    dur = defaultdict(lambda: [0, 0.0])
    ctr = defaultdict(lambda: defaultdict(float))
    nd = defaultdict(set)
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = rd.fieldnames or []
            if "Start_Timestamp" in cols and "Kernel_Name" in cols and "Counter_Name" not in cols:
                for r in rd:
                    k = short(r["Kernel_Name"])
                    dur[k][0] += 1
                    dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            elif "Counter_Name" in cols:
                for r in rd:
                    k = short(r["Kernel_Name"])
                    ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    nd[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    if dur:
        tot = sum(v[1] for v in dur.values())
        print(f"# kernel trace: total kernel time {tot:.3f} ms over {sum(v[0] for v in dur.values())} launches", file=out)
        print(f"{'kernel':112s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'share':>6s}", file=out)
        for k, (n, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:112s} {n:6d} {t:10.3f} {t / n:9.4f} {100 * t / tot:5.1f}%", file=out)
    if ctr:
        print("\n# counters: mean per launch (summed over all counter dimensions)", file=out)
        for k in sorted(ctr, key=lambda k: -sum(ctr[k].values())):
            n = max(len(nd[k]), 1)
            vals = "  ".join(f"{c}={v / n:.4g}" for c, v in sorted(ctr[k].items()))
            print(f"{k:112s} n={n:5d}  {vals}", file=out)


if __name__ == "__main__":
    main()
