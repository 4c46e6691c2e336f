set -e
mkdir -p gpurun_out/r3p
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_reads or ragged or golden or random or overlapped" > gpurun_out/r3p/pytest.txt 2>&1 || { tail -30 gpurun_out/r3p/pytest.txt; exit 1; }
tail -2 gpurun_out/r3p/pytest.txt
B="python bench.py --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0"
$B > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err
$B --workload small > gpurun_out/r3p/bench_small.json 2> gpurun_out/r3p/bench_small.err
for rps in 131072 262144; do $B --steps 20 --warmup 3 --reads-per-step $rps > gpurun_out/r3p/greedy_$rps.json 2> gpurun_out/r3p/greedy_$rps.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3p/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], [(k["kernel"][:24], k["avg_ms"]) for k in d["roofline"]["kernels_ms"]], d["parity_sample"])
PY
