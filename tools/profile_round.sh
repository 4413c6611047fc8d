#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag>   -- kernel-trace stats of the default bench + PMC traffic
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $O/step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_pl -- python3 $R/bench.py --no-cpu-baseline --no-extras --powerlaw --steps 20 --warmup 3 > $O/step_pl.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_sparse -- python3 $R/bench.py --no-cpu-baseline --no-extras --mode sparse --steps 20 --warmup 3 > $O/step_sparse.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/spmm -- python3 $R/bench.py --only-spmm > $O/spmm.log 2>&1
for d in step step_pl step_sparse spmm; do f=$(ls $O/$d/*/*kernel_stats.csv | head -1); cp $f $O/${d}_kernel_stats.csv; echo "== $d"; head -12 $f; done
cd $R && tools/pmc_traffic.sh 21
