"""Record the PMC-measured HBM traffic of the dominant kernel for bench.py's `roofline.traffic`.

    python tools/update_conv_traffic.py profiles/rNN_pmc_fetch.summary.txt profiles/rNN_pmc_write.summary.txt [--batch 8]

Reads the per-launch means of FETCH_SIZE / WRITE_SIZE (KiB; tools/prof_summary.py output of two separate rocprofv3 --pmc
passes of `python bench.py ...`) for the dominant kernel (md_conv3_wino_kernel; --kernel to override), applies the gfx950 correction of MI355X_MICROARCH.md
(FETCH_SIZE reports half of a wide streaming read: doubled), and stores bytes per launch in profiles/conv_traffic.json
under the sha of the kernel's source files -- bench.py only reports the figure while that sha still matches.
"""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNEL = "md_conv3_wino_kernel"


def counter(path, name, form):
    """Launch-weighted mean of counter `name` over the summary lines of KERNEL whose template arguments start with `form`
    (round 6: the kernel exists in a residual and a no-residual form per arithmetic: "<0, true, true" = both f16f6 forms)."""
    tot = n = 0.0
    for ln in open(path):
        if KERNEL + form in ln and f"{name}=" in ln:
            k = float(re.search(r"n=\s*(\d+)", ln).group(1))
            tot += k * float(re.search(name + r"=([0-9.e+]+)", ln).group(1))
            n += k
    if not n:
        raise SystemExit(f"{name} of {KERNEL}{form} not found in {path}")
    return tot / n


def main():
    global KERNEL
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--kernel", default=KERNEL, help="substring of the kernel name in the summaries")
    ap.add_argument("--source", default="conv3_wino.hip", help="source file whose sha keys the entry")
    ap.add_argument("--precision", default="f16f6", help="hip_precision of the profiled run (bench.py reports the entry only for the same one)")
    ap.add_argument("--form", default=None, help="leading template arguments of the kernel forms to average over (default: by --precision)")
    a = ap.parse_args()
    KERNEL = a.kernel
    form = a.form if a.form is not None else {"f16f6": "<0, true, true", "f16f8": "<0, true, false", "bf16x3": "<0, false, false"}[a.precision]
    fetch_kib, write_kib = counter(a.fetch, "FETCH_SIZE", form), counter(a.write, "WRITE_SIZE", form)
    nbytes = (2.0 * fetch_kib + write_kib) * 1024.0
    try:
        with open(bench.TRAFFIC_FILE) as fh:
            tr = json.load(fh)
    except OSError:
        tr = {}
    key = bench.conv_source_key(a.source)
    tr[key] = {"kernel": f"{KERNEL}{form}, ...> (the build bench.py runs by default; launch-weighted over its residual / no-residual forms)", "hbm_bytes_per_launch": round(nbytes), "batch": a.batch, "precision": a.precision,
               "fetch_size_kib": fetch_kib, "write_size_kib": write_kib,
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB, mean over the launches of `python bench.py`",
               "source": f"{os.path.relpath(a.fetch, ROOT)} + {os.path.relpath(a.write, ROOT)}"}
    with open(bench.TRAFFIC_FILE, "w") as fh:
        json.dump(tr, fh, indent=1, sort_keys=True)
    print(f"{key}: {nbytes / 1e9:.3f} GB per launch")


if __name__ == "__main__":
    main()
