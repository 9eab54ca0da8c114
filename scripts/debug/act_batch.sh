#!/bin/bash
# acting-kernel measurements in one gpurun call: launch times over shapes / wave shapes / experiment builds, in-kernel stage timing, HBM traffic by PMC
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_act; mkdir -p $O; cd $R
{
for epw in 0 8 4; do MAGPO_ACT_EPW=$epw python scripts/debug/act_time.py 16384 4 1; done
for epw in 0 8 16; do MAGPO_ACT_EPW=$epw python scripts/debug/act_time.py 4096 4 1; done
for epw in 4 8 16; do MAGPO_ACT_EPW=$epw python scripts/debug/act_time.py 8192 4 1; done
for epw in 4 8; do MAGPO_ACT_EPW=$epw python scripts/debug/act_time.py 2048 4 1; done
for epw in 8 16; do MAGPO_ACT_EPW=$epw python scripts/debug/act_time.py 16384 8 2 15 12; done
python scripts/debug/act_time.py 16384 8 2 15 12
python scripts/debug/act_time.py 16384 2 1 6
for lib in "$@"; do MAGPO_LIB=$lib python scripts/debug/act_time.py 16384 4 1; done
} 2>&1 | grep -v "amdgpu.ids" | tee $O/times.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/scripts/debug/act_time.py 16384 4 1 20 20 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/scripts/debug/act_time.py 16384 4 1 20 20 > $O/pmc_write.log 2>&1
python3 $R/scripts/pmc_collect.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_act.json | grep -i "sable_act" | tee -a $O/times.txt
