#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3j; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "sinc or load_audio or end_to_end or resident or pcm16" 2>&1 | grep "passed\|failed" 
python tools/exp_sinc.py 2>&1 | grep -v amdgpu | tee $O/sinc.log
BATCHES=128,128 NO_PROFILE=1 python tools/exp_e2e_profile.py 2>&1 | grep batch_files | tee $O/e2e.log
