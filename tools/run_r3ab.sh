set -e
mkdir -p gpurun_out/r3ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_reads or ragged or golden or random or overlapped" > gpurun_out/r3ab/pytest.txt 2>&1 || { tail -30 gpurun_out/r3ab/pytest.txt; exit 1; }
tail -2 gpurun_out/r3ab/pytest.txt
B="python bench.py --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0"
$B > gpurun_out/r3ab/ecoli.json 2> gpurun_out/r3ab/ecoli.err
$B --workload small > gpurun_out/r3ab/small.json 2> gpurun_out/r3ab/small.err
$B --steps 20 --warmup 3 --reads-per-step 131072 > gpurun_out/r3ab/g131k.json 2> gpurun_out/r3ab/g131k.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3ab/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], [(k["kernel"][:24], k["avg_ms"]) for k in d["roofline"]["kernels_ms"]], d["parity_sample"]["gpu_equals_oracle"], d["counters"])
PY
