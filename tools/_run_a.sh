set -x
timeout 600 python -m pytest tests/test_gpu_ois.py -x -q 2>&1 | tail -5
timeout 600 python bench.py 2>&1 | tail -3
