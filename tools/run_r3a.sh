set -e
mkdir -p gpurun_out/r3a
rocprofv3 -L > gpurun_out/r3a/counters.txt 2>&1 || true
tools/ubench/valu_rates > gpurun_out/r3a/valu_rates.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_reads or ragged or golden or random or overlapped" > gpurun_out/r3a/pytest.txt 2>&1 || { tail -30 gpurun_out/r3a/pytest.txt; exit 1; }
tail -3 gpurun_out/r3a/pytest.txt
B="python bench.py --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0"
$B > gpurun_out/r3a/bench_o8.json 2> gpurun_out/r3a/bench_o8.err
BGR_LIB_PATH=$PWD/bgreat_amd/lib_o7/libbgreat_gpu.so $B > gpurun_out/r3a/bench_o7.json 2> gpurun_out/r3a/bench_o7.err
BGR_LIB_PATH=$PWD/bgreat_amd/lib_o6/libbgreat_gpu.so $B > gpurun_out/r3a/bench_o6.json 2> gpurun_out/r3a/bench_o6.err
$B --workload small > gpurun_out/r3a/bench_small.json 2> gpurun_out/r3a/bench_small.err
python - <<'PY'
import json
for t in ("o8","o7","o6","small"):
    d=json.load(open("gpurun_out/r3a/bench_%s.json"%t))
    print(t, d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"], d["config"]["launch"], d["config"]["pass_counts_last_launch"], d["parity_sample"])
PY
