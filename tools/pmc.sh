#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" -- collects PMC counters for bench.py kernels (separate pass per call)
tag=$1; ctrs=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-pmc --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v)/len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
