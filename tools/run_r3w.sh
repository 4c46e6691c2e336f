set -e
mkdir -p gpurun_out/r3w
timeout -k 10 900 python -m pytest tests/test_gpu_text.py tests/test_gpu_parity.py -x -q -m gpu -k "text or correction or cli or asynchronous" > gpurun_out/r3w/pytest.txt 2>&1 || { tail -40 gpurun_out/r3w/pytest.txt; exit 1; }
tail -3 gpurun_out/r3w/pytest.txt
