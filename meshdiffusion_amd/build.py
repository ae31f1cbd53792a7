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
OBJDIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libmeshdiffusion_hip.so")
ARCH = "gfx950"
SOURCES = ["capi.hip", "gemm_conv.hip", "conv3_main.hip", "norm.hip", "elementwise.hip", "train.hip", "backward.hip", "wgrad.hip", "dmtet.hip"]
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
    """Compile every HIP source for gfx950 and link the shared library. Returns its path."""
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "md_common.h"), os.path.join(INCLUDE, "meshdiffusion_hip.h")]
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise RuntimeError(f"missing source {sp}")
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _newer(obj, [sp] + headers):
            cmd = [hipcc] + FLAGS + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(obj)
    if force or _newer(LIB_PATH, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
