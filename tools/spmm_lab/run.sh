#!/bin/bash
# usage (on the GPU box, from the repo root): tools/spmm_lab/run.sh <tag> <scale> "<variants for the PMC passes>"
tag=$1; scale=$2; pmcv=$3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab_$tag; mkdir -p $O
LAB=$R/tools/spmm_lab/spmm_lab
$LAB $scale 16 256 10 > $O/time_s$scale.txt 2>&1
cat $O/time_s$scale.txt
if [ -n "$pmcv" ]; then
  cd /tmp && export TMPDIR=/tmp
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_32B_sum"; do
    name=$(echo $ctr | tr ' ' '_')
    rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_s${scale}_$name -- $LAB $scale 16 256 2 "$pmcv" > $O/pmc_s${scale}_$name.log 2>&1
  done
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$O/pmc_s${scale}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/pmc_s${scale}_summary.txt", "w") as f:
    for k, d in sorted(acc.items()):
        line = k + " " + str({c: round(sum(v) / len(v), 1) for c, v in d.items()}) + " n=" + str(len(next(iter(d.values()))))
        print(line); f.write(line + "\n")
PY
fi
