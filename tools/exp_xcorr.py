"""Developer experiment: time ssr_xcorr_argmax on 4 s @ 48 kHz pairs (N4)."""
import os, sys, json, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B, _lib

def main():
    n_items, n = int(os.environ.get("ITEMS", "64")), 192000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    x = 0.1 * torch.randn((n_items, n), generator=g, device=dev)
    y = torch.roll(x, 1105, dims=1) + 0.01 * torch.randn((n_items, n), generator=g, device=dev)
    ra, rb = B.Ragged.from_uniform(y.contiguous()), B.Ragged.from_uniform(x.contiguous())
    lib = _lib.load()
    ws_bytes = int(lib.ssr_xcorr_workspace_bytes(ra.n, ra.max_len))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty(ra.n, dtype=torch.int64, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr())
    run = lambda: _lib.check(lib.ssr_xcorr_argmax(vp(ra.data), vp(ra.off), vp(rb.data), vp(rb.off), vp(ra.len), ra.n, ra.max_len,
                                                  vp(out), vp(ws), ws_bytes, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ms = bench.event_time_ms(run, 3)
    macs = n_items * float(n) * n
    print(json.dumps({"items": n_items, "ms": round(ms, 3), "ms_per_item": round(ms / n_items, 4),
                      "TMAC_per_s": round(macs / (ms * 1e-3) / 1e12, 2), "argmax_minus_n": int(out[0].item()) - n}))

if __name__ == "__main__":
    main()
