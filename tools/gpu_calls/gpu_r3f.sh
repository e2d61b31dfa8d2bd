#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3f; mkdir -p $O
BATCHES=32,64,128,367 NO_PROFILE=1 python tools/exp_e2e_profile.py 2>&1 | tail -6 | tee $O/e2e.log
BATCHES=64 python tools/exp_e2e_profile.py 2>&1 | tail -30 | tee -a $O/e2e.log
