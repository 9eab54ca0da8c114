#!/bin/bash
# A/B of retention-kernel builds on ONE box: parity tests, launch times (twice, interleaved) and the LDS counters of each library.
# usage: bash scripts/debug/ret32_ab.sh <lib.so> ...   (magpo_amd/libmagpo_hip.so is always included)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ret32_ab; mkdir -p $O
LIBS="magpo_amd/libmagpo_hip.so $@"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "retention_chunk" > $O/tests.log 2>&1; echo "parity tests (magpo_amd/libmagpo_hip.so): $(tail -1 $O/tests.log)"
for rep in 1 2; do
  for lib in $LIBS; do
    echo "== $(basename $lib .so) (run $rep)"
    MAGPO_LIB=$lib RET_ONLY32=1 timeout -k 10 300 python scripts/debug/ret32_time.py 2>&1 | grep -v amdgpu.ids
  done
done
for lib in $LIBS; do
  n=$(basename $lib .so)
  export MAGPO_LIB=$GRAFT_REPO_ROOT/$lib RET_ONLY32=1 RET_REPS=1
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$n -- python3 $GRAFT_REPO_ROOT/scripts/debug/ret32_time.py > $GRAFT_REPO_ROOT/$O/pmc_$n.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 scripts/pmc_sq_summary.py $(find $O/pmc_$n -name "*counter_collection.csv" | head -1) $O/lds_counters_$n.csv > /dev/null && grep ret32 $O/lds_counters_$n.csv | sed "s/^/$n: /"
  head -1 $O/lds_counters_$n.csv
  rm -rf $O/pmc_$n
done
