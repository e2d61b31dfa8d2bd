#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for L in "" tools/_build/libssrhip_lpw_nostore.so "" tools/_build/libssrhip_lpw_nostore.so; do SSR_DEV_LIB=$L _ONE=1 python tools/exp_lowpass.py 2>/dev/null | tail -1; done | tee gpurun_out/r4an_lpw.log
