#!/bin/bash
# quick perf check: ER + power-law, all modes (GPU box)
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --spmm-scale 0 --steps 30 > /tmp/quick_pl.json
python - <<'PY'
import json
d = json.loads(open('/tmp/quick_pl.json').read().strip().splitlines()[-1])
print('ER full ms', d['ms_per_step'], ' kernels', {k: v['avg_us'] for k, v in d['kernels'].items()})
print('ER other', {k: v['ms_per_step'] for k, v in d['other_modes'].items()})
print('PL', d['workload_2']['ms_per_step'], d['workload_2']['sparse_ms_per_step'])
PY
