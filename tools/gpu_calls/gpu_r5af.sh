#!/bin/bash
# round 5 call af: fp64_mix with the exchanges' DS instructions split (2 x ds_read_b64 / 2 x ds_write_b64 against the read2 / write2 forms)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 120 tools/_build/fp64_mix 2>&1 | tee gpurun_out/r5af_fp64_mix.log
