"""Developer experiment: overlap k_ssim of sub-batch i with k_stft of sub-batch i+1 on two streams."""
import os, sys, json, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B

def main():
    dev = torch.device("cuda", 0)
    n = 1024
    est, tgt = bench.make_inputs(n, dev, 1)
    plan = B.get_plan(2048, 512, "f64", dev)
    mask = B.M_LSD | B.M_SSIM
    whole = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
    print("serial 1024:", round(bench.event_time_ms(lambda: whole.run(mask), 10), 3), "ms")
    for nsub in (2, 8, 16, 32, 64):
        k = n // nsub
        subs = [B.PairBatch(plan, B.Ragged.from_uniform(est[i * k:(i + 1) * k].contiguous()),
                            B.Ragged.from_uniform(tgt[i * k:(i + 1) * k].contiguous())) for i in range(nsub)]
        s_main = torch.cuda.current_stream()
        s2 = torch.cuda.Stream()
        def run():
            evs = []
            for sb in subs:
                sb.run(mask, stages=1)                       # STFT on the main stream
                ev = torch.cuda.Event(); ev.record(s_main); evs.append(ev)
                with torch.cuda.stream(s2):
                    s2.wait_event(ev)
                    sb.run(mask, stages=6)                   # SSIM + finalize on the side stream
            done = torch.cuda.Event(); done.record(s2); s_main.wait_event(done)
        ms = bench.event_time_ms(run, 10)
        def serial():
            for sb in subs:
                sb.run(mask)
        ms_serial = bench.event_time_ms(serial, 10)
        ms_ssim = bench.event_time_ms(lambda: [sb.run(mask, stages=2) for sb in subs], 10)
        print("%d slices of %d pairs (%.0f MB of magnitudes each): pipelined on two streams %.3f ms, same slices on one stream %.3f ms, their k_ssim launches alone %.3f ms"
              % (nsub, k, k * 376 * 1028 * 8 / 1e6, ms, ms_serial, ms_ssim))
        torch.cuda.synchronize()
        ref = whole.run(mask).clone()
        got = torch.cat([sb.out for sb in subs])
        print("   max rel diff vs serial:", float(((got - ref).abs() / ref.abs()).nan_to_num().max()))

if __name__ == "__main__":
    main()
