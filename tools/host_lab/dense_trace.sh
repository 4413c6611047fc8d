cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr; LT_FEATURE_DELTA=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/bench.py --steps 20 --no-extras --no-api-wall --no-pmc --no-cpu-baseline --blocks 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[1:12]:
    print("  ", r[0][:60], r[1], round(float(r[3])/1e3,1), 'us avg')
f=glob.glob('/tmp/tr/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[(r["Kernel_Name"][:36], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, int(r["Start_Timestamp"])) for r in rows]
mid=len(names)//2
prev=None
for n,d,t in names[mid:mid+16]:
    print(f"{d:8.1f}  gap {((t-prev)/1e3 if prev else 0):6.1f}  {n}")
    prev=t+int(d*1e3)
PY
