#!/bin/bash
# round 6 call f: micro-variants of the PAIRED k_stft_wave (bins in flight, scheduling barriers, request points) and the launch
# geometry (interleave group, workgroups per launch) on one box, alternating with the shipped build; SQ counters of k_stft_r3_rot
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { SSR_DEV_LIB=$1 NO_CHECK=1 timeout 200 python tools/exp_stage.py 2>&1 | tail -1 | cut -c1-230; }
for rep in 1 2; do
  for L in "" tools/_build/libssrhip_g4.so tools/_build/libssrhip_nosb.so tools/_build/libssrhip_pf1.so; do run "$L"; done
done | tee gpurun_out/r6f_variants.log
K=tools/_build/libssrhip_knobs.so
for S in 1 4 8 16; do echo "interleave $S"; SSR_WAVE_INTERLEAVE=$S run $K; done | tee gpurun_out/r6f_interleave.log
for W in 2048 4096 8192 16384; do echo "target wgs $W"; SSR_TARGET_WGS=$W run $K; done | tee gpurun_out/r6f_wgs.log
tools/pmc_cmd.sh r6f_api k_stft_r3_rot -- python tools/exp_api_true.py 2>&1 | tail -30 | tee gpurun_out/r6f_r3rot_sq.txt
