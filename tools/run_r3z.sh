set -e
mkdir -p gpurun_out/r3z
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3z/pytest.txt 2>&1 || { tail -40 gpurun_out/r3z/pytest.txt; exit 1; }
tail -3 gpurun_out/r3z/pytest.txt
B="python bench.py --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0"
$B > gpurun_out/r3z/ecoli.json 2> gpurun_out/r3z/ecoli.err
$B --workload chr1 > gpurun_out/r3z/chr1.json 2> gpurun_out/r3z/chr1.err
$B --workload branchy > gpurun_out/r3z/branchy.json 2> gpurun_out/r3z/branchy.err
$B --workload small > gpurun_out/r3z/small.json 2> gpurun_out/r3z/small.err
$B --anchors > gpurun_out/r3z/anchors.json 2> gpurun_out/r3z/anchors.err
$B --exhaustive > gpurun_out/r3z/exh.json 2> gpurun_out/r3z/exh.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3z/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], [(k["kernel"][:24], k["avg_ms"]) for k in d["roofline"]["kernels_ms"]], d["parity_sample"]["gpu_equals_oracle"])
PY
grep "graph:" gpurun_out/r3z/chr1.err gpurun_out/r3z/ecoli.err gpurun_out/r3z/branchy.err | cut -c1-400
