import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ssr_eval_amd import backend as B
dev = torch.device('cuda', 0)
N, n = 20000, 48000
g = torch.Generator(device=dev).manual_seed(1)
tgt = 0.1 * torch.randn((N, n), generator=g, device=dev)
est = (tgt + 0.01 * torch.randn((N, n), generator=g, device=dev)).contiguous()
plan = B.get_plan(2048, 512, 'f64', dev)
b = B.PairBatch(plan, B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
torch.cuda.synchronize(); t0 = time.time()
out = b.run().cpu().numpy(); dt = time.time() - t0
print('20000 x 1 s pairs: %.1f ms -> %.0f pairs/s; ws %.1f GB' % (dt * 1e3, N / dt, b.ws_bytes / 1e9))
print('means', out.mean(0), 'nan', np.isnan(out).sum(), 'std lsd', out[:, 0].std())
# ragged: 3000 items with lengths U(1.5, 9) s
rng = np.random.default_rng(0)
lens = rng.integers(72000, 432000, 3000)
flat = 0.1 * torch.randn(int(lens.sum()), generator=g, device=dev)
noise = 0.01 * torch.randn(int(lens.sum()), generator=g, device=dev)
off = np.concatenate(([0], np.cumsum(lens)[:-1]))
def rag(x):
    return B.Ragged(x, torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(lens.astype(np.int32)).to(dev), lens)
pb = B.PairBatch(plan, rag((flat + noise).contiguous()), rag(flat))
torch.cuda.synchronize(); t0 = time.time(); o2 = pb.run().cpu().numpy(); dt = time.time() - t0
print('2937-like ragged (3000 utt, %.0f s audio): %.1f ms -> %.0f utt/s' % (lens.sum() / 48000, dt * 1e3, 3000 / dt), o2.mean(0), np.isnan(o2).sum())
