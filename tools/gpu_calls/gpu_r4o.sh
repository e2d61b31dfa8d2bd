#!/bin/bash
# round 4: is k_resample_rc memory-bound?  timing-only builds: every workgroup reads item 0's input / writes item 0's output
for L in "" tools/_build/libssrhip_samein.so tools/_build/libssrhip_sameout.so tools/_build/libssrhip_sameio.so; do
  echo "lib=$L"; SSR_DEV_LIB=$L NO_MFMA=1 python tools/exp_resample.py 2>&1 | tail -1
done
