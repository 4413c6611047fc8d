#!/bin/bash
# usage: tools/prof_one.sh <tag> <bench args...>   -- rocprofv3 kernel stats of one bench configuration (GPU box)
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-pmc --steps 20 --warmup 3 "$@" > $O.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/*/*kernel_stats.csv")[0]
print("== $tag")
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.3:
        print(r['Name'].split('(')[0][:60].ljust(62), r['Calls'].rjust(4), '%9.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
