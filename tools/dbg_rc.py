import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
sys.path.insert(0,"/root/repo/tools"); import devlib; devlib.select()
from scipy import signal
from ssr_eval_amd import backend as B
rng=np.random.default_rng(0)
for up,down,n in ((441,160,6400),(160,147,8820),(441,160,64000),(160,147,30000)):
    x=(0.1*rng.standard_normal(n)).astype(np.float32)
    y=B.resample_poly([x, x[:n//2+7]], up, down)
    for xi,yi in zip((x, x[:n//2+7]), y):
        ref=signal.resample_poly(xi, up, down); got=yi.cpu().numpy()
        bad=np.nonzero(got!=ref)[0]
        print(up,down,len(xi),len(ref),'mismatch',len(bad), bad[:10], bad[-5:] if len(bad) else '', np.abs(got-ref).max())
