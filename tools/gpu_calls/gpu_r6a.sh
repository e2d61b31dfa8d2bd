#!/bin/bash
# round 6 call a: the premise of VERDICT r5 item 1 - clock + SQ counters of the SHIPPED k_stft_wave / k_ssim in the same session as
# the fp64_mix ceiling; phase stamps of the shipped frame loop; a bench line of this box.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$PWD
timeout 120 tools/_build/fp64_mix > gpurun_out/r6a_fp64_mix.log 2>&1; echo "ubench rc=$?"; cat gpurun_out/r6a_fp64_mix.log
OUT=$R/gpurun_out/pmc_r6a; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  NO_CHECK=1 PAIRS=1024 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o c -- python $R/tools/exp_stage.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
{ echo "== fp64_mix, same session"; cat gpurun_out/r6a_fp64_mix.log
  echo "== clock (GRBM_GUI_ACTIVE / dispatch duration)"; python tools/pmc_clock.py $OUT/p1
  echo "== SQ counters, medians per launch (tools/exp_stage.py, 1024 pairs)"; python tools/pmc_summary.py $OUT k_stft k_ssim; } > gpurun_out/r6a_stft_wave_sq.txt 2>&1
cat gpurun_out/r6a_stft_wave_sq.txt
find $OUT -name "*.csv" -size +1M -delete
echo "== phase stamps"
SSR_DEV_LIB=tools/_build/libssrhip_clocks.so NO_CHECK=1 timeout 200 python tools/exp_stage.py > gpurun_out/r6a_clocks.log 2>&1; tail -40 gpurun_out/r6a_clocks.log | cut -c1-400
echo "== bench"
timeout 400 python bench.py > gpurun_out/r6a_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r6a_bench.log | cut -c1-1500
echo "== occupancy experiment: the float32 instance of the SAME frame loop at 2 (shipped, 176 VGPRs) / 3 (168, 29 spilled) / 4 (128, 95 spilled) waves per SIMD"
for L in "" tools/_build/libssrhip_f32w3.so tools/_build/libssrhip_f32w4.so; do
  for rep in 1 2; do SSR_DEV_LIB=$L PRECS=f32 NO_CHECK=1 timeout 200 python tools/exp_stage.py 2>&1 | tail -1; done
done | tee gpurun_out/r6a_f32_occupancy.log
