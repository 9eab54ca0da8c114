#!/bin/bash
# Round-2 measurement pass on the GPU box (one gpurun call): bench lines of every workload, rocprofv3 kernel statistics of the
# default bench command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) behind profiles/r02_pmc_hbm_traffic.json.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err && tail -c 600 $O/bench_default.json
python bench.py --num-envs 4096 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_4096envs.json 2> $O/bench_4096envs.err && cut -c1-330 $O/bench_4096envs.json
python bench.py --workload coordsum-8x15 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_8x15.json 2> $O/bench_8x15.err && cut -c1-330 $O/bench_8x15.json
python bench.py --workload lbf-8x8-2p-2f --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_lbf.json 2> $O/bench_lbf.err && cut -c1-330 $O/bench_lbf.json
python bench.py --workload rware-tiny-4ag --num-envs 4096 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rware.json 2> $O/bench_rware.err && cut -c1-330 $O/bench_rware.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --steps 3 > $O/prof_default.json 2> $O/prof_default.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/scripts/pmc_kernels.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/scripts/pmc_kernels.py > $O/pmc_write.log 2>&1
ls $O $O/pmc_fetch/* | head -40
