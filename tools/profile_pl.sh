#!/bin/bash
# usage (GPU box, repo root): tools/profile_pl.sh <tag>  -- rocprofv3 kernel-trace stats of the power-law delta step alone
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 3 --blocks 1 --powerlaw"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_pl -- python3 $R/bench.py $B > $O/step_pl.log 2>&1
f=$(ls $O/step_pl/*/*kernel_stats.csv | head -1); cp $f $O/step_pl_kernel_stats.csv; head -8 $f | cut -d, -f1-4 | cut -c1-110
