set -e
mkdir -p gpurun_out/r3al
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3al/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r3al/gpu_tests.log; exit 1; }
tail -2 gpurun_out/r3al/gpu_tests.log
