#!/bin/bash
# A/B of an environment knob on ONE box: tools/ab_env.sh <rounds> VAR=a VAR=b ...   (interleaved rounds)
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline --no-extras --steps 20 $LT_AB_ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'step_ms', d['ms_per_step'], 'stageA_us', d['kernels'].get('full_stageA',{}).get('avg_us'))"
  done
done
