#!/bin/bash
# round 5 call j: forward product with the 4 x 1 wave layout for a last tile of <= 64 bins - parity, timing, conv-related tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
UTT=1024 timeout 600 python tools/exp_tlconv.py > gpurun_out/r5j_tlconv.log 2>&1
echo "rc=$?"; grep -v "^stft\|^uniform" gpurun_out/r5j_tlconv.log | tail -28
timeout 600 python -m pytest tests -m gpu -q --timeout 240 -k "lowpass or conv or fdomain or cfg3 or multi or cfg1 or waveform" > gpurun_out/r5j_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r5j_tests.log
