#!/bin/bash
# tools/ab3.sh <rounds> so1 so2 ... : all with LT_STAGEA_LDS=1
rounds=$1; shift
cp linkteller_amd/liblinkteller_hip.so /tmp/_orig.so
for r in $(seq $rounds); do for so in "$@"; do cp $so linkteller_amd/liblinkteller_hip.so
 LT_STAGEA_LDS=1 python bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'step_ms', d['ms_per_step'], 'stageA_us', d['kernels']['full_stageA']['avg_us'])"
done; done; cp /tmp/_orig.so linkteller_amd/liblinkteller_hip.so
