#!/bin/bash
# HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md "HBM") of the kernels bench.py prices, from separate
# rocprofv3 --pmc passes of bench.py itself.  Run on the GPU box from the repo root:  tools/pmc_traffic.sh [spmm_scale]
# Writes gpurun_out/pmc_traffic.json (copy to profiles/pmc_traffic.json: bench.py reads it when the kernel sources match).
scale=${1:-21}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_traffic; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $O/step_$ctr -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-pmc --steps 3 --warmup 1 > $O/step_$ctr.log 2>&1
  rocprofv3 --pmc $ctr --output-format csv -d $O/spmm_$ctr -- python3 $R/bench.py --only-spmm --no-pmc --spmm-scale $scale > $O/spmm_$ctr.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections, subprocess, sys
sys.path.insert(0, "$R")
import bench
def collect(prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob("$O/%s_*/*/*counter_collection.csv" % prefix):
        for r in csv.DictReader(open(fn)):
            acc[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
def hbm(d):   # KiB counters; FETCH_SIZE reports half of the bytes of wide reads on gfx950
    return int((2 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024)
step, spmm = collect("step"), collect("spmm")
out = {"source_signature": bench.source_signature(), "unit": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
       "command": "tools/pmc_traffic.sh $scale (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py)",
       "kernels": {}, "raw": {"step": step, "spmm": spmm}}
if "k_full_stageA_lds" in step:
    out["kernels"]["full_stageA"] = {"hbm_bytes_per_launch": hbm(step["k_full_stageA_lds"])}
for k in ("k_gemm_f32_mfma_128",):
    if k in step:
        out["kernels"]["gemm"] = {"hbm_bytes_per_launch": hbm(step[k])}
tot = sum(hbm(spmm[k]) for k in ("k_rows_tiled", "k_spmm_long_combine") if k in spmm)
if tot:
    out["kernels"]["spmm_rmat$scale"] = {"hbm_bytes_per_launch": tot, "parts": {k: hbm(spmm[k]) for k in ("k_rows_tiled", "k_spmm_long_combine") if k in spmm}}
json.dump(out, open("$R/gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
