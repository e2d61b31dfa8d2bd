#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/pc$v
  (cd $R && SSR_NO_FUSED_OLA=$v SSR_DEV_LIB=tools/_build/libssrhip_knobs.so timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc$v -o p -- python tools/exp_cfg3_seq.py) > /tmp/pc$v.log 2>&1
  echo "== SSR_NO_FUSED_OLA=$v"; tail -1 /tmp/pc$v.log | cut -c1-300; python $R/tools/pmc_clock.py /tmp/pc$v | head -6
done
