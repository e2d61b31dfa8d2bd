#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
BATCHES=64,128,128 NO_PROFILE=1 STAGES=1 python tools/exp_e2e_profile.py 2>&1 | tail -5 | tee $O/e2e.log
