#!/bin/bash
# round 6 call aa: what the packed IIR loop's vector-memory instructions cost - developer builds without the regular chunk's stores / loads / both
# (their outputs are wrong by construction: timings only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for L in "" tools/_build/libssrhip_iirnostore.so tools/_build/libssrhip_iirnoload.so tools/_build/libssrhip_iirnoio.so; do
  SSR_DEV_LIB=$L PER_G=1 FILES=64 timeout 300 python tools/exp_iir.py 2>&1 | tail -1 | cut -c1-700
done | tee gpurun_out/r6aa_iir_io_cost.log
