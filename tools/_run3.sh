export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_decode_wide.py -x -q 2>&1 | tail -15
timeout 200 python tools/wide_stamps.py 256 2>&1 | grep "^pro"
timeout 200 python tools/wide_stamps.py 768 2>&1 | grep "^pro"
for b in 256 768; do for wm in 0 32; do AUDIOCAPTION_DEC_WIDE_MIN=$wm timeout 200 python tools/decode_wide_bench.py $b 2>&1 | tail -1; done; done
