#!/bin/bash
# usage: tools/prof_rmat.sh <tag> <scale> <n_probes> <modes> <n_observed>  -- rocprofv3 kernel stats of tools/rmat_check.py (GPU box)
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/tools/rmat_check.py "$@" > $O.log 2>&1
cat $O.log | grep -v "^W\|rocprof" | tail -12
python3 - <<PY
import csv,glob
f=glob.glob("$O/*/*kernel_stats.csv")[0]
print("== $tag")
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.5:
        print(r['Name'].split('(')[0][:60].ljust(62), r['Calls'].rjust(4), '%9.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
