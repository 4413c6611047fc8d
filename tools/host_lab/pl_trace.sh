cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 1; do
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/bench.py --steps 20 --no-extras --powerlaw > /dev/null 2>&1
  echo "== power-law step"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[1:8]:
    print(r[0][:44], r[1], round(float(r[3])/1e3,1), 'min', round(float(r[5])/1e3,1), 'max', round(float(r[6])/1e3,1))
PY
done
