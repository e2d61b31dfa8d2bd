#!/bin/bash
# round 6 call p: what a step of the packed IIR wavefront waits for - clock, s_waitcnt share, VALU busy, instructions per step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$PWD
OUT=$R/gpurun_out/pmc_r6p; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  FILES=64 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o c -- python $R/tools/exp_iir.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
cd $R
{ python tools/pmc_clock.py $OUT/p1; python tools/pmc_summary.py $OUT k_sosfiltfilt; } 2>&1 | tee gpurun_out/r6p_iir_sq.txt
find $OUT -name "*.csv" -size +1M -delete
