#!/bin/bash
# round 4 call ao: SQ counters of k_lowpass_wave (1024 x 4 s, cut 256) - where the waves' cycles go
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
_ONE=1 bash tools/pmc_cmd.sh lpw k_lowpass_wave -- python tools/exp_lowpass.py 2>&1 | tail -20 | tee gpurun_out/r4ao_lpw_sq.log
find gpurun_out/pmc_lpw -name "*.csv" -size +200k -delete
