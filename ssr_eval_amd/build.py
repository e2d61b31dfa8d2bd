"""Build libssrhip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "ssr_hip.hip")
OUT = os.path.join(_HERE, "libssrhip.so")


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = glob.glob(os.path.join(_HERE, "csrc", "*")) + [os.path.join(os.path.dirname(_HERE), "include", "ssr_hip.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
