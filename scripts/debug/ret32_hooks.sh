#!/bin/bash
# launch times of the retention kernels under the timing hooks of retention32.hpp + the in-kernel phase profile
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in magpo_amd/libmagpo_hip.so "$@"; do
    echo "== $(basename $lib .so) (run $rep)"
    MAGPO_LIB=$lib RET_ONLY32=1 timeout -k 10 300 python scripts/debug/ret32_time.py 2>&1 | grep -v amdgpu.ids
  done
done
MAGPO_LIB=exp_libs/ret_prof.so timeout -k 10 300 python scripts/ret_prof.py 2>&1 | grep -v amdgpu.ids
