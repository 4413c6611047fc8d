cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in delta sparse; do
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/tools/host_lab/gcn3_trace.py $m 2>&1 | tail -1
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[1:14]:
    print("  ", r[0][:70], r[1], round(float(r[3])/1e3,1), 'us avg', round(float(r[2])/35e3,1), 'us per step')
PY
done
