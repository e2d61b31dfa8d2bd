#!/bin/bash
# round 4: SQ counters of the residue-class resampler
NO_MFMA=1 ITERS=2 bash tools/pmc_cmd.sh rc k_resample_rc -- python tools/exp_resample.py 2>&1 | tail -40
