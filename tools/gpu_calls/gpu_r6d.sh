#!/bin/bash
# round 6 call d: zero-scratch PAIRED variants against the ascending order; phase stamps of the PAIRED frame loop; the API-true and
# float64-estimate paths after the late prefetch (timing + their parity tests); round-6 profile set with PMC
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2 3; do
  for L in "" tools/_build/libssrhip_asc.so; do SSR_DEV_LIB=$L NO_CHECK=1 timeout 200 python tools/exp_stage.py 2>&1 | tail -1; done
done | tee gpurun_out/r6d_paired_ab.log
SSR_DEV_LIB=tools/_build/libssrhip_clocks.so NO_CHECK=1 timeout 200 python tools/exp_stage.py > gpurun_out/r6d_clocks.log 2>&1
for k in "sums=0 mag=1" "sums=0 mag=0" "sums=1 mag=1"; do grep "$k" gpurun_out/r6d_clocks.log | tail -2; done | cut -c1-330 | tee gpurun_out/r6d_phase_stamps_paired.log
for L in "" tools/_build/libssrhip_asc.so; do SSR_DEV_LIB=$L timeout 300 python tools/exp_api_true.py 2>&1 | tail -2; done | tee gpurun_out/r6d_api_true.log
IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | tail -2 | cut -c1-600 | tee gpurun_out/r6d_e2e_iir.log
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -k "float64 or est64 or rates or api or r3 or 2229 or iir or sos" 2>&1 | tail -4 | tee gpurun_out/r6d_tests.log
PMC=1 timeout 2400 bash tools/collect_profiles_r06.sh r06 2>&1 | tail -60
