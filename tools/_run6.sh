export TMPDIR=/tmp PYTHONPATH=$PWD AUDIOCAPTION_TRAFFIC_OPTIONAL=1
timeout 600 python -m pytest tests/test_gpu_decode_wide.py tests/test_gpu_model.py -x -q -k "hybrid or beam or general_launch" 2>&1 | tail -4
for hy in 0 1 0 1; do
  AUDIOCAPTION_DEC_HYBRID=$hy timeout 300 python bench.py --mode effb2 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hybrid=$hy effb2', d['value'], d['ms_per_step'])"
done
for hy in 0 1; do
  AUDIOCAPTION_DEC_HYBRID=$hy timeout 300 python bench.py --mode effb2 --seconds 30 --beam 4 --effb2-batch 64 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hybrid=$hy effb2 30s beam4', d['value'], d['ms_per_step'])"
done
for ntb in 2 4 8; do AUDIOCAPTION_DEC_WIDE_CLS_NTB=$ntb timeout 300 python bench.py --mode effb2 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cls ntb=$ntb effb2', d['value'], d['ms_per_step'])"; done
