#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
python tools/exp_sinc.py 2>&1 | tee $O/sinc.log
NO_PROFILE=1 STAGES=1 python tools/exp_e2e_profile.py 2>&1 | tail -8 | tee $O/e2e.log
