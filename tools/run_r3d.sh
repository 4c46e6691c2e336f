set -e
mkdir -p gpurun_out/r3d
timeout -k 10 600 python -m pytest tests/test_gpu_text.py -x -q -m gpu > gpurun_out/r3d/pytest_text.txt 2>&1 || { tail -40 gpurun_out/r3d/pytest_text.txt; exit 1; }
tail -3 gpurun_out/r3d/pytest_text.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cli" > gpurun_out/r3d/pytest_cli.txt 2>&1 || { tail -40 gpurun_out/r3d/pytest_cli.txt; exit 1; }
tail -3 gpurun_out/r3d/pytest_cli.txt
BGREAT_TIMING=1 python tools/e2e.py --reads 40000000 --check 300000 > gpurun_out/r3d/e2e.json 2> gpurun_out/r3d/e2e.err || { tail -20 gpurun_out/r3d/e2e.err; exit 1; }
tail -12 gpurun_out/r3d/e2e.err; cat gpurun_out/r3d/e2e.json
