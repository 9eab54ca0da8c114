#!/bin/bash
# Round-3 measurement pass on the GPU box (one gpurun call): kernel statistics of the default bench command by rocprofv3.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --steps 3 > $O/prof_default.json 2> $O/prof_default.err
f=$(find $O/prof_default -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -32 $O/kernel_stats.csv | cut -c1-200
