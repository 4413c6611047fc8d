#!/bin/bash
# observed-hub stage B at twitch size (power-law graph): short-side search on / off / automatic, sparse and delta (GPU box)
cd $GRAFT_REPO_ROOT
for m in sparse delta; do for h in 0 1 auto; do
  if [ $h = auto ]; then unset LT_HUB_SHORT_SIDE; else export LT_HUB_SHORT_SIDE=$h; fi
  python bench.py --no-cpu-baseline --no-extras --powerlaw --mode $m --steps 30 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PL $m hub_short_side=$h', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done; done
