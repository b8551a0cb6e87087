export TMPDIR=/tmp PYTHONPATH=$PWD
Q="--no-cpu-baseline --no-tiers --no-train --no-effb2 --no-ingest --no-ragged --no-steady-state --steps 40"
for lib in "" tools/bin/libdeccap64.so "" tools/bin/libdeccap64.so; do
  AUDIOCAPTION_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-shipped} headline', d['value'], d['ms_per_step'])"
done
for lib in "" tools/bin/libdeccap64.so; do
  AUDIOCAPTION_HIP_LIB=${lib:+$PWD/$lib} timeout 200 python tools/decode_wide_bench.py 256 2>&1 | tail -1
done
