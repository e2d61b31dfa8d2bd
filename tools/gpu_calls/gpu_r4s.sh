#!/bin/bash
# round 4 call s: k_specred_wave keys per wave after packing (knobs build) + kernel stats of the multi call
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SSR_DEV_LIB=tools/_build/libssrhip_knobs.so
for kg in 1 2 3; do SSR_SPEC_KG=$kg python tools/exp_multi.py 2>&1 | tail -1; done | tee gpurun_out/r4s_kg.log
unset SSR_DEV_LIB
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4s_prof -o s -- python $R/tools/exp_multi.py > $R/gpurun_out/r4s_prof.log 2>&1
cd $R
f=$(find gpurun_out/r4s_prof -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
