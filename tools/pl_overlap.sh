cd $GRAFT_REPO_ROOT
for o in 1 0 1 0; do
  LT_OVERLAP=$o python bench.py --no-cpu-baseline --no-extras --powerlaw --steps 30 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PL full overlap=$o', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
