#!/bin/bash
# round 6 call h: kernel-level breakdown of an evaluate() pass with the 36 IIR keys
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r6h_prof
BATCH_FILES=64 PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6h_prof -o s -- python $R/tools/exp_e2e.py > $R/gpurun_out/r6h_e2e.log 2>&1
cd $R; grep "evaluate()" gpurun_out/r6h_e2e.log | cut -c1-200
head -14 gpurun_out/r6h_prof/*kernel_stats.csv | cut -c1-200 | tee gpurun_out/r6h_kernel_stats_head.csv
find gpurun_out/r6h_prof -name "*kernel_trace.csv" -delete
