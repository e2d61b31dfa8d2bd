"""Developer tooling only: run a tool against an alternative build of the library (A/B of kernel variants, -DSSR_DEV_KNOBS
builds).  The product binding (ssr_eval_amd/_lib.py) loads ssr_eval_amd/libssrhip.so and nothing else; tools that want
another build call `devlib.select()` before the first library call, with SSR_DEV_LIB=<path to the .so> in the environment.
Alternative builds live under tools/_build/ (git-ignored; they travel to the GPU box with the snapshot):

    python tools/devlib.py build knobs -DSSR_DEV_KNOBS            -> tools/_build/libssrhip_knobs.so
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BUILD_DIR = os.path.join(ROOT, "tools", "_build")


def select():
    path = os.environ.get("SSR_DEV_LIB")
    if not path:
        return None
    from ssr_eval_amd import _lib
    if _lib._lib is not None:
        raise RuntimeError("devlib.select() must run before the first library call")
    path = path if os.path.isabs(path) else os.path.join(ROOT, path)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    _lib.LIB_PATH = path
    # an older build (the baseline of an A/B) may lack entry points added since: the tools bind what it has
    import ctypes
    probe = ctypes.CDLL(path)
    for name in [k for k in _lib.SIGNATURES if not hasattr(probe, k)]:
        del _lib.SIGNATURES[name]
    return path


def build(tag, flags):
    """Compile every unit with extra flags into tools/_build/libssrhip_<tag>.so (objects under tools/_build/obj_<tag>/)."""
    from ssr_eval_amd import build as b
    os.makedirs(BUILD_DIR, exist_ok=True)
    b.OBJ = os.path.join(BUILD_DIR, "obj_" + tag)
    return b.build(force=False, verbose=False, extra=list(flags), out=os.path.join(BUILD_DIR, "libssrhip_%s.so" % tag))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "build":
        print(build(sys.argv[2], sys.argv[3:]))
    else:
        print(__doc__)


def build_ref(ref, tag):
    """Build the library as of git revision `ref` (sources extracted to a scratch directory) into
    tools/_build/libssrhip_<tag>.so - the baseline of a same-box A/B."""
    import shutil
    import subprocess
    import tempfile
    from ssr_eval_amd import build as b
    tmp = tempfile.mkdtemp(prefix="ssr_ref_")
    try:
        subprocess.check_call("git -C %s archive %s ssr_eval_amd/csrc include | tar -x -C %s" % (ROOT, ref, tmp), shell=True)
        os.makedirs(BUILD_DIR, exist_ok=True)
        b.CSRC = os.path.join(tmp, "ssr_eval_amd", "csrc")
        b.OBJ = os.path.join(tmp, "obj")
        return b.build(force=True, verbose=False, out=os.path.join(BUILD_DIR, "libssrhip_%s.so" % tag))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
