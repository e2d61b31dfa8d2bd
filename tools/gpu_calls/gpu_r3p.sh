#!/bin/bash
for i in 1 2 3; do
SSR_DEV_LIB=tools/_build/libssrhip_r02.so python tools/exp_resample.py 2>&1 | tail -1 | sed 's/^/r02: 1-D, 44 unconditional prefetch loads: /'
SSR_DEV_LIB=tools/_build/libssrhip_rs1d.so python tools/exp_resample.py 2>&1 | tail -1 | sed 's/^/1-D, prefetch in guarded groups:          /'
python tools/exp_resample.py 2>&1 | tail -1 | sed 's/^/2-D, prefetch in guarded groups:          /'
done
