#!/bin/bash
# round 6 call y: the randomized differential sweeps of earlier rounds on the head of round 6 (nothing else changed them: a regression check)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for T in stress_parity stress_degrade stress_sinc stress_r04 stress_r05 stress_iir; do
  echo "== $T"; timeout 900 python tools/$T.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
done | tee gpurun_out/r6y_stress_all.log
