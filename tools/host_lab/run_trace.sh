cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 16 8 0" "30 16 8 15" "60 16 8 15" "60 16 8 40" "45 16 8 25"; do
  set -- $cfg
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/tools/host_lab/host_trace.py $1 $2 $3 $4 > /dev/null 2>&1
  echo "== share $1 waves $2 inflight $3 share2 $4"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[1:5]:
    print(r[0][:40], r[1], round(float(r[3])/1e3,1), 'min', round(float(r[5])/1e3,1), 'max', round(float(r[6])/1e3,1))
PY
done
