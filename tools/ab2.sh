#!/bin/bash
# tools/ab2.sh <rounds> "<so> <P>" ...  : interleaved A/B of (library variant, LT_FULL_P) pairs on one box
rounds=$1; shift
cfgs=("$@")
cp linkteller_amd/liblinkteller_hip.so /tmp/_orig.so
for r in $(seq $rounds); do
  for cfg in "${cfgs[@]}"; do
    so=${cfg% *}; P=${cfg##* }
    cp $so linkteller_amd/liblinkteller_hip.so
    LT_FULL_P=$P python bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so P=$P', 'step_ms', d['ms_per_step'], 'stageA_us', d['kernels'].get('full_stageA',{}).get('avg_us'))"
  done
done
cp /tmp/_orig.so linkteller_amd/liblinkteller_hip.so
