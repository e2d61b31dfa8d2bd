#!/bin/bash
# k_ssim: y ring in LDS + wave-scope phases (default build) against HEAD~ (libssrhip_ssim0.so): kernel alone, the whole step, values
mkdir -p gpurun_out/r3q
for i in 1 2 3; do
LIBS=tools/_build/libssrhip_ssim0.so, python tools/exp_ssim.py 2>&1 | grep '^{'
done | tee gpurun_out/r3q/ssim_ab.txt
python -m pytest tests -m gpu -x -q -k "ssim or cfg2 or golden or spectrogram" 2>&1 | tail -5 | tee gpurun_out/r3q/pytest.txt
python bench.py 2>/dev/null | tail -1 | tee gpurun_out/r3q/bench.json
