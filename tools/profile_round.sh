#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag>   -- rocprofv3 kernel-trace stats of the bench configurations
# (default = --mode delta; the dense-feature route of delta; full; sparse; power-law delta and full; the R-MAT SpMM leg)
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 3 --blocks 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $R/bench.py $B > $O/step.log 2>&1
LT_FEATURE_DELTA=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_dense -- python3 $R/bench.py $B > $O/step_dense.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_full -- python3 $R/bench.py $B --mode full > $O/step_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_sparse -- python3 $R/bench.py $B --mode sparse > $O/step_sparse.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_pl -- python3 $R/bench.py $B --powerlaw > $O/step_pl.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_pl_full -- python3 $R/bench.py $B --powerlaw --mode full > $O/step_pl_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/spmm -- python3 $R/bench.py --only-spmm --no-pmc > $O/spmm.log 2>&1
for d in step step_dense step_full step_sparse step_pl step_pl_full spmm; do f=$(ls $O/$d/*/*kernel_stats.csv | head -1); cp $f $O/${d}_kernel_stats.csv; echo "== $d"; head -9 $f | cut -c1-160; tail -2 $O/$d.log | cut -c1-300; done
