#!/bin/bash
# usage (GPU box, repo root): tools/pmc_spmm.sh <tag> [scale]
# L2 hit / miss and memory-side read counters of the R-MAT SpMM kernel (k_rows_tiled) AND of its gather ceiling
# (k_rows_tiled_gathers_only), one rocprofv3 --pmc pass per counter set over `bench.py --only-spmm` (the hit
# structure the ceiling argument rests on, from counters instead of a host-side column histogram).
tag=$1; scale=${2:-21}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_spmm_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "TCC_[A-Za-z0-9_]*" $O/avail.txt | sort -u > $O/tcc_counters.txt
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RD_UNCACHED_32B_sum" \
           "TCC_REQ_sum TCC_READ_sum" "TCC_BUBBLE_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '+')
  rocprofv3 --pmc $set --output-format csv -d $O/$name -- python3 $R/bench.py --only-spmm --no-pmc --spmm-scale $scale > $O/$name.log 2>&1
  echo "$name rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0][:48]
        if k.startswith("k_rows_tiled") or k.startswith("k_spmm_long"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(acc.items()):
    out[k] = {c: round(sum(v) / len(v), 1) for c, v in d.items()}
    out[k]["dispatches"] = len(next(iter(d.values())))
    h, m = out[k].get("TCC_HIT_sum"), out[k].get("TCC_MISS_sum")
    if h is not None and m is not None and h + m > 0:
        out[k]["l2_hit_frac"] = round(h / (h + m), 4)
    print(k, out[k])
json.dump(out, open("$O/summary.json", "w"), indent=1)
PY
