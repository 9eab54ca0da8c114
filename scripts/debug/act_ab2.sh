#!/bin/bash
# Elimination experiments on the acting kernel (results of the X builds are WRONG; only their time matters).  Builds, on the CPU box:
#   for v in NOROWMATH NOW NOMFMA NOSAMPLE; do python -m magpo_amd.build --out exp_libs/x_$v.so --only act_fused --flags=-DMAGPO_X_$v; done
# then on the GPU box: bash scripts/debug/act_ab2.sh exp_libs/x_*.so
cd $GRAFT_REPO_ROOT
for lib in magpo_amd/libmagpo_hip.so "$@"; do
  MAGPO_LIB=$lib python scripts/debug/act_time.py 16384 4 1 2>&1 | grep -v amdgpu.ids
done
