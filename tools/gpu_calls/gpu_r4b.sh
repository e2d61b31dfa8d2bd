#!/bin/bash
# round 4, call 2: the full GPU suite with the conv engine as the default of lowpass(_type="stft_hard") / FDomainHelper
python -c "from oracle import tl_chain; tl_chain.build()"
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40
cp -f gpurun_out/r04_cfg3_engines.json gpurun_out/ 2>/dev/null; ls gpurun_out | head
