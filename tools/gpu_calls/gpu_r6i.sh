#!/bin/bash
# round 6 call i: evaluate() with the 36 IIR keys - more (design, utterance) recurrences per IIR launch (the launch is as long as its
# longest utterance: 576 waves on 1024 SIMDs at 64 files x 36 designs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for CFG in "64 29" "128 29" "128 31" "256 31" "256 32" "367 32"; do
  set -- $CFG
  echo "batch_files $1, output doubles per launch 2^$2"
  SSR_SOS_MULTI_MAX_DOUBLES=$((1 << $2)) BATCH_FILES=$1 PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-200
done | tee gpurun_out/r6i_iir_launch_size.log
