"""Build libssrhip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

The library is several translation units (ssr_eval_amd/csrc/tu_*.hip: one per kernel family / transform precision),
compiled in parallel to objects under csrc/_obj/ and linked into one shared object; only units whose sources changed
are recompiled.
"""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
OUT = os.path.join(_HERE, "libssrhip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def units():
    return sorted(glob.glob(os.path.join(CSRC, "tu_*.hip")))


def _headers_mtime():
    deps = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        [os.path.join(os.path.dirname(_HERE), "include", "ssr_hip.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps)


def _obj_of(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(src, hdr_mtime, extra):
    o = _obj_of(src)
    stamp = o + ".flags"
    if not os.path.exists(o) or not os.path.exists(stamp) or open(stamp).read() != " ".join(extra):
        return True
    return max(os.path.getmtime(src), hdr_mtime) > os.path.getmtime(o)


def needs_build(extra=()):
    if not os.path.exists(OUT):
        return True
    h = _headers_mtime()
    return any(_stale(u, h, list(extra)) for u in units()) or any(
        os.path.getmtime(_obj_of(u)) > os.path.getmtime(OUT) for u in units())


def build(force=False, verbose=False, extra=(), jobs=None, out=None):
    """extra: additional compiler flags (e.g. -DSSR_DEV_KNOBS for the profiling tools)."""
    extra = list(extra)
    out = out or OUT
    if not force and out == OUT and not needs_build(extra):
        return out
    os.makedirs(OBJ, exist_ok=True)
    cc, h = hipcc(), _headers_mtime()
    todo = [u for u in units() if force or _stale(u, h, extra)]

    def one(src):
        cmd = [cc] + FLAGS + extra + ["-c", src, "-o", _obj_of(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        open(_obj_of(src) + ".flags", "w").write(" ".join(extra))

    with ThreadPoolExecutor(max_workers=jobs or max(1, (os.cpu_count() or 2))) as ex:
        list(ex.map(one, todo))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [_obj_of(u) for u in units()] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, extra=[a for a in sys.argv[1:] if a.startswith("-D")]))
