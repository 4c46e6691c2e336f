set -e
mkdir -p gpurun_out/r3m
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3m/pytest.txt 2>&1 || { tail -40 gpurun_out/r3m/pytest.txt; exit 1; }
tail -3 gpurun_out/r3m/pytest.txt
