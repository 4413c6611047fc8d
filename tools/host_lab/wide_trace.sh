cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/tools/host_lab/wide_trace.py $cfg 2>&1 | grep "ms per step\|Baseline"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[1:12]:
    print("  ", r[0][:66], r[1], round(float(r[3])/1e3,1), 'us avg', round(float(r[2])/23e3,1), 'us per step')
PY
done
