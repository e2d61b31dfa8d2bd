#!/bin/bash
# round 4, call 3: FLAC ingest e2e + the Ragged aliasing fix
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "flac or ragged_from_list or wav_files or mp3" 2>&1 | tail -30
