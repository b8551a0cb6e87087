export TMPDIR=/tmp PYTHONPATH=$PWD
Q="--no-cpu-baseline --no-tiers --no-train --no-effb2 --no-ingest --no-ragged --no-steady-state --steps 40"
for wm in 0 192 0 192; do
  AUDIOCAPTION_DEC_WIDE_MIN=$wm timeout 300 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide_min=$wm headline', d['value'], d['ms_per_step'])"
done
for wm in 0 192 0 192; do
  AUDIOCAPTION_DEC_WIDE_MIN=$wm timeout 300 python bench.py --mode effb2 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide_min=$wm effb2', d['value'], d['ms_per_step'])"
done
