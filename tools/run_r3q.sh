set -e
mkdir -p gpurun_out/r3q
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3q/pytest.txt 2>&1 || { tail -30 gpurun_out/r3q/pytest.txt; exit 1; }
tail -2 gpurun_out/r3q/pytest.txt
B="python bench.py --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0"
$B --anchors > gpurun_out/r3q/anchors.json 2> gpurun_out/r3q/anchors.err
$B --exhaustive > gpurun_out/r3q/exh.json 2> gpurun_out/r3q/exh.err
$B --workload branchy > gpurun_out/r3q/branchy.json 2> gpurun_out/r3q/branchy.err
$B --workload chr1 > gpurun_out/r3q/chr1.json 2> gpurun_out/r3q/chr1.err
for rps in 131072 262144; do $B --steps 20 --warmup 3 --reads-per-step $rps --anchors > gpurun_out/r3q/anchors_$rps.json 2> gpurun_out/r3q/anchors_$rps.err; $B --steps 20 --warmup 3 --reads-per-step $rps --exhaustive > gpurun_out/r3q/exh_$rps.json 2> gpurun_out/r3q/exh_$rps.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3q/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], [(k["kernel"][:24], k["avg_ms"]) for k in d["roofline"]["kernels_ms"]], d["parity_sample"])
PY
