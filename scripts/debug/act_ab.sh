#!/bin/bash
# A/B of acting-kernel builds on ONE box: each library timed twice, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in magpo_amd/libmagpo_hip.so "$@"; do
    MAGPO_LIB=$lib python scripts/debug/act_time.py 16384 4 1 2>&1 | grep -v amdgpu.ids
    MAGPO_LIB=$lib MAGPO_ACT_EPW=8 python scripts/debug/act_time.py 4096 4 1 2>&1 | grep -v amdgpu.ids
  done
done
