export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_greedy_cluster.py tests/test_gpu_decode_wide.py -x -q 2>&1 | tail -15
