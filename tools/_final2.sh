set -e -o pipefail
python tools/ws_stress.py 60 2>&1 | tail -2
timeout -k 10 700 python -m pytest tests/test_ws_gpu.py tests/test_kernels_gpu.py tests/test_tape_gpu.py tests/test_headline_size_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/_final.sh f8b3327
