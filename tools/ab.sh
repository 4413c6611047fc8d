#!/bin/bash
# A/B on ONE box: tools/ab.sh <rounds> variantA.so variantB.so ...   (interleaved rounds; prints stage-A avg us)
rounds=$1; shift
cp linkteller_amd/liblinkteller_hip.so /tmp/_orig.so
for r in $(seq $rounds); do
  for v in "$@"; do
    cp $v linkteller_amd/liblinkteller_hip.so
    python bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'step_ms', d['ms_per_step'], 'stageA_us', d['kernels'].get('full_stageA',{}).get('avg_us'), 'gemm', d['kernels']['gemm']['avg_us'])"
  done
done
cp /tmp/_orig.so linkteller_amd/liblinkteller_hip.so
