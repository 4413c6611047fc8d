cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr; export LT_SHARD_TRACE=1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/tools/shard_time.py 21 ${1:-delta} 2>&1 | grep -i "error" ; unset LT_SHARD_TRACE
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[1:26]:
    print("  ", r[0][:66], r[1], round(float(r[3])/1e3,1), 'us avg', round(float(r[2])/10e3,1), 'us per build')
PY
