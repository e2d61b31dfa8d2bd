#!/bin/bash
# Run ON THE GPU BOX: SQ counters of an arbitrary command (developer tool).  usage: tools/pmc_cmd.sh <tag> <kernel-key> -- cmd...
TAG=$1; KEY=$2; shift 3
R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o c -- "$@") > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python $R/tools/pmc_summary.py $OUT $KEY
