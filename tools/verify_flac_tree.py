#!/usr/bin/env python
"""Validate a tree of FLAC files with the library's own decoder before handing it to SSR_Eval_Helper.evaluate().

    python tools/verify_flac_tree.py <dir> [--quiet]

Every *.flac under <dir> is decoded through the C ABI (ssr_flac_info / ssr_flac_decode_i32: the native decoder of
ssr_eval_amd/csrc/ssr_flac.h, no GPU needed) with its STREAMINFO MD5 checked against the decoded PCM - the decoder refuses a
stream whose samples do not hash to the encoder's own digest, so a file that passes here was decoded exactly as its encoder
(libFLAC for VCTK 0.92) meant it.  Prints rate / channels / bits / frames per file, a summary, and every failure with the
decoder's message; exit status 1 if any file failed, 2 if no FLAC file was found.

Why it exists: the decoder and the test encoder (tests/flac_fixture.py) were written from RFC 9639 by
the same hand and the image holds no libFLAC-written stream; a holder of the VCTK test set (Zenodo 6370601,
ssr_eval/eval.py:106) runs this ONE command first.  A clean run plus evaluate() then reproduces README.md:34-41 of the reference
except for the two resamplers that are not in its tree - see "Reproducing the reference's published numbers" in README.md.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("root")
    ap.add_argument("--quiet", action="store_true", help="print failures and the summary only")
    a = ap.parse_args(argv)
    from ssr_eval_amd import _lib
    import numpy as np
    lib = _lib.load()
    files = sorted(os.path.join(d, f) for d, _, fs in os.walk(a.root) for f in fs if f.lower().endswith(".flac"))
    if not files:
        print("no .flac file under %s" % a.root)
        return 2
    bad, n_frames, t0 = [], 0, time.time()
    seen = {}
    for path in files:
        sr, nch, bits, md5 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        frames = C.c_int64()
        rc = lib.ssr_flac_info(path.encode(), C.byref(sr), C.byref(nch), C.byref(bits), C.byref(frames), C.byref(md5))
        if rc == 0 and frames.value == 0:              # a streamed file without a sample count in STREAMINFO: count first
            rc = lib.ssr_flac_decode_i32(path.encode(), None, 0, 0, C.byref(frames))
        if rc == 0:
            buf = np.empty(max(int(frames.value) * int(nch.value), 1), dtype=np.int32)
            got = C.c_int64()
            rc = lib.ssr_flac_decode_i32(path.encode(), buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size), 1, C.byref(got))
            if rc == 0 and got.value != frames.value:
                rc, msg = -1, "decoded %d of %d frames" % (got.value, frames.value)
        if rc != 0:
            msg = lib.ssr_last_error()
            bad.append((path, msg.decode() if msg else "error %d" % rc))
            print("FAIL %s: %s" % (path, bad[-1][1]))
            continue
        key = (sr.value, nch.value, bits.value)
        seen[key] = seen.get(key, 0) + 1
        n_frames += frames.value
        if not md5.value:
            print("NOTE %s: STREAMINFO carries no MD5 (all zero): decoded, but not verifiable" % path)
        if not a.quiet:
            print("ok   %s: %d Hz, %d ch, %d bit, %d frames%s" % (path, sr.value, nch.value, bits.value, frames.value,
                                                                "" if md5.value else " (no MD5)"))
    dt = time.time() - t0
    print("%d file(s), %d failed, %d frames decoded and MD5-checked in %.1f s; formats (rate, channels, bits): %s" % (
        len(files), len(bad), n_frames, dt, ", ".join("%s x%d" % (k, v) for k, v in sorted(seen.items()))))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
