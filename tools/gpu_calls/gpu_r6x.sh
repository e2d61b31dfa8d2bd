#!/bin/bash
# round 6 call x: k_stft_wave with the magnitude rows as streaming (nt) stores - A/B, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do
  for L in "" tools/_build/libssrhip_rowsnt.so; do SSR_DEV_LIB=$L NO_CHECK=1 timeout 200 python tools/exp_stage.py 2>&1 | tail -1 | cut -c1-330; done
done | tee gpurun_out/r6x_rows_nt_ab.log
