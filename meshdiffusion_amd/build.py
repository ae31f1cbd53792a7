"""Build libmeshdiffusion_hip.so (gfx950) in-tree with hipcc.

`python -m meshdiffusion_amd.build` or `meshdiffusion_amd.build.build()`.
hipcc cross-compiles for gfx950 without a GPU; objects are cached by mtime.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
# A/B builds of a kernel under development (tools/bench_wino.py --lib): MD_LIB_SUFFIX=_v1 MD_EXTRA_DEFINES="-DW8_SCHED=1" writes
# libmeshdiffusion_hip_v1.so from its own object directory; MD_LIB=<path> makes _lib.py load it.  The default build has neither.
_SUFFIX = os.environ.get("MD_LIB_SUFFIX", "")
OBJDIR = os.path.join(CSRC, "build" + _SUFFIX)
LIB_PATH = os.path.join(HERE, f"libmeshdiffusion_hip{_SUFFIX}.so")
ARCH = "gfx950"
SOURCES = ["capi.hip", "gemm_conv.hip", "conv3_main.hip", "conv3_wino.hip", "conv3_s2.hip", "conv3_head.hip", "conv3_stem.hip", "pack_batch.hip", "wino_prep2.hip", "wino_eq.hip", "norm.hip", "elementwise.hip", "attention.hip", "nin_stream.hip", "train.hip", "backward.hip", "wgrad.hip", "wgrad_wino.hip", "dmtet.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", f"-I{INCLUDE}", f"-I{CSRC}",
         "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libmeshdiffusion_hip.so")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False, force=False):
    """Compile every HIP source for gfx950 (in parallel) and link the shared library. Returns its path.
    MD_BUILD_ABLATIONS=1 in the environment also builds the timing-only kernel variants of tools/bench_conv.py /
    tools/bench_wino.py (adds ~4 minutes: seven more instantiations of the unrolled 27-tap conv kernel)."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "md_common.h"), os.path.join(CSRC, "md_pack.h"), os.path.join(INCLUDE, "meshdiffusion_hip.h")]
    flags = FLAGS + (["-DMD_BUILD_ABLATIONS"] if os.environ.get("MD_BUILD_ABLATIONS") == "1" else [])
    flags = flags + os.environ.get("MD_EXTRA_DEFINES", "").split()
    stamp = os.path.join(OBJDIR, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise RuntimeError(f"missing source {sp}")
        obj = os.path.join(OBJDIR, os.path.basename(src).replace(".hip", ".o"))
        if force or _newer(obj, [sp] + headers):
            jobs.append([hipcc] + flags + ["-c", sp, "-o", obj])
        objs.append(obj)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
        with open(stamp, "w") as f:
            f.write(" ".join(flags))
        force = True
    if force or _newer(LIB_PATH, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
