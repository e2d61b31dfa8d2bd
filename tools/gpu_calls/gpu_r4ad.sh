#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time python __graft_entry__.py smoke ) 2>&1 | grep -v "hipcc\|^$" | tail -8
