#!/bin/bash
python tools/dbg_rc.py 2>&1 | tail -8
for i in 1 2; do NO_MFMA=1 python tools/exp_resample.py 2>&1 | tail -1; done
NO_MFMA=1 UTT=12500 ITERS=3 python tools/exp_resample.py 2>&1 | tail -1
NO_MFMA=1 ITERS=2 bash tools/pmc_cmd.sh rc k_resample_rc -- python tools/exp_resample.py 2>&1 | tail -17
