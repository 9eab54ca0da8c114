#!/bin/bash
# Round-4 measurement pass on the GPU box (one gpurun call): bench lines of every workload, rocprofv3 kernel statistics of the default bench
# command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) behind profiles/r04_pmc_hbm_traffic.json.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err && cut -c1-420 $O/bench_default.json
python bench.py --num-envs 4096 --steps 20 --warmup 3 --no-cpu-baseline --no-variants > $O/bench_4096envs.json 2> $O/bench_4096envs.err && cut -c1-330 $O/bench_4096envs.json
python bench.py --workload coordsum-8x15 --steps 5 --warmup 2 --no-cpu-baseline --no-variants > $O/bench_8x15.json 2> $O/bench_8x15.err && cut -c1-330 $O/bench_8x15.json
python bench.py --workload lbf-8x8-2p-2f --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $O/bench_lbf.json 2> $O/bench_lbf.err && cut -c1-330 $O/bench_lbf.json
python bench.py --workload rware-tiny-4ag --num-envs 4096 --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $O/bench_rware.json 2> $O/bench_rware.err && cut -c1-330 $O/bench_rware.json
python bench.py --workload rware-tiny-4ag --num-envs 4096 --embed-dim 128 --n-head 2 --n-block 3 --ppo-epochs 5 --steps 4 --warmup 1 --no-cpu-baseline --no-variants > $O/bench_rware_tuned_e128.json 2> $O/bench_rware_tuned_e128.err && cut -c1-330 $O/bench_rware_tuned_e128.json
python bench.py --gpus 2 --backend gloo --num-envs 4096 --steps 3 --warmup 1 --check-replicas --no-cpu-baseline > $O/bench_2rank_gloo_one_gpu.json 2> $O/bench_2rank_gloo_one_gpu.err && cut -c1-330 $O/bench_2rank_gloo_one_gpu.json
for e in 4 8 16; do MAGPO_ACT_EPW=$e python scripts/debug/act_time.py 16384 2>&1 | grep -v amdgpu.ids >> $O/act_time.txt; MAGPO_ACT_EPW=$e python scripts/debug/act_time.py 4096 2>&1 | grep -v amdgpu.ids >> $O/act_time.txt; done
python scripts/debug/act_time.py 8192 2>&1 | grep -v amdgpu.ids >> $O/act_time.txt
python scripts/debug/act_time.py 16384 8 2 15 2>&1 | grep -v amdgpu.ids >> $O/act_time.txt
python scripts/debug/act_time.py 16384 2 1 6 2>&1 | grep -v amdgpu.ids >> $O/act_time.txt
cat $O/act_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --no-variants --steps 3 > $O/prof_default.json 2> $O/prof_default.err
cp $(find $O/prof_default -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/scripts/pmc_kernels.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/scripts/pmc_kernels.py > $O/pmc_write.log 2>&1
python3 $R/scripts/pmc_collect.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm_traffic.json | head -14
rm -rf $O/pmc_fetch $O/pmc_write $O/prof_default
