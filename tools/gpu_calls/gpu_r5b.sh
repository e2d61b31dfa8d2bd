#!/bin/bash
# round 5 call b: k_tl_inv without the per-chunk accumulator copies (one loop body); variants; per-kernel times under rocprofv3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
R=$PWD
SSR_DEV_LIB=tools/_build/libssrhip_knobs.so VARIANTS="128:2,128:3,64:2" UTT=1024 timeout 900 python tools/exp_tlconv.py > gpurun_out/r5b_tlconv.log 2>&1
echo "rc=$?"; grep -v "^per-item\|^uniform\|^stft" gpurun_out/r5b_tlconv.log | tail -45
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r5b_stats
PARITY=0 UTT=1024 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5b_stats -o s -- python $R/tools/exp_tlconv.py > $R/gpurun_out/r5b_stats.log 2>&1
echo "stats rc=$?"
find $R/gpurun_out/r5b_stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -20
find $R/gpurun_out/r5b_stats -name "*kernel_trace.csv" -delete; find $R/gpurun_out/r5b_stats -name "*agent_info.csv" -delete
