#!/bin/bash
# Developer tool: register / spill / scratch figures of the kernels of ONE translation unit whose mangled name contains a key.
#   tools/ru.sh tu_stft_f64_p0 k_stft_wave [extra hipcc flags]
U=$1; K=$2; shift 2
cd "$(dirname "$0")/../ssr_eval_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $U.hip -o /tmp/ru_$U.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -A14 "Function Name: .*$K" | grep -E "Function Name|VGPRs:|Spill|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - - - - | sed 's/Function Name: //' | c++filt | cut -c1-250
