#!/bin/bash
# power-law `full` step: hub-row kernels on the side stream (default) / on the caller's stream (LT_OVERLAP=0), hub rows as
# segments in separate waves (LT_LONG_PAR=1, default at this size) / one wave per (row, 16 probes) (LT_LONG_PAR=0)   (GPU box)
cd $GRAFT_REPO_ROOT
for cfg in "1 1" "1 0" "0 1" "0 0" "1 1" "1 0"; do
  set -- $cfg
  LT_OVERLAP=$1 LT_LONG_PAR=$2 python bench.py --no-cpu-baseline --no-extras --powerlaw --steps 30 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PL full overlap=$1 long_par=$2', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
