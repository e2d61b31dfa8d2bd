#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for p in f64 f32 f64 f32; do PREC=$p _ONE=1 python tools/exp_lowpass.py; done | tee gpurun_out/r4af_lowpass_prec.log
