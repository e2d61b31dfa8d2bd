#!/bin/bash
# round 6 call l: a CLEAN occupancy experiment - the float32 instance of the paired frame loop WITHOUT frame-ahead requests
# (-DSSR_WAVE_NOPF: 170 VGPRs at two waves per SIMD, 168 and NO spill at three), against the shipped float32 instance (prefetch, two waves)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do
  for L in "" tools/_build/libssrhip_nopf2.so tools/_build/libssrhip_nopf3.so; do SSR_DEV_LIB=$L PRECS=f32 NO_CHECK=1 timeout 200 python tools/exp_stage.py 2>&1 | tail -1 | cut -c1-230; done
done | tee gpurun_out/r6l_f32_nopf_occupancy.log
