#!/bin/bash
# Development: headline of bench.py with the decode chains on a CU-masked stream (AUDIOCAPTION_DECODE_CUS), a few settings.
F="--no-cpu-baseline --no-tiers --no-effb2 --no-ingest --no-ragged --no-train --steps 40"
for v in ${SWEEP:-0 16 32 64 96 "32,exclusive" "16,exclusive"}; do
  echo "== AUDIOCAPTION_DECODE_CUS=$v"
  AUDIOCAPTION_DECODE_CUS="$v" timeout 300 python bench.py $F 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','steady_state_value','value_blocking_model_call')})"
done
