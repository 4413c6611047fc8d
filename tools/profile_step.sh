tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 3 --blocks 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $R/bench.py $B > $O/step.log 2>&1
f=$(ls $O/step/*/*kernel_stats.csv | head -1); head -6 $f | cut -c1-150; tail -1 $O/step.log | cut -c1-200
