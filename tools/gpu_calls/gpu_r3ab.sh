#!/bin/bash
# larger randomized sweeps over every engine after the round's kernel changes (k_ssim, k_stft_r3_rot, SSIM geometry)
mkdir -p gpurun_out/r3ab
for s in 501 502 503 504 505 506 507 508; do SEED=$s CASES=100 timeout 900 python tools/stress_parity.py 2>&1 | tail -1; done | tee gpurun_out/r3ab/stress_parity.txt
for s in 601 602 603 604; do SEED=$s CASES=40 timeout 900 python tools/stress_degrade.py 2>&1 | tail -1; done | tee gpurun_out/r3ab/stress_degrade.txt
PREC=f32 SEED=701 CASES=60 timeout 600 python tools/stress_parity.py 2>&1 | tail -1 | tee gpurun_out/r3ab/stress_f32.txt
