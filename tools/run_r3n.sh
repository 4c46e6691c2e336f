set -e
mkdir -p gpurun_out/r3n
bash tools/phase_cost.sh > gpurun_out/r3n/phase_cost.txt 2>&1 || { tail -20 gpurun_out/r3n/phase_cost.txt; exit 1; }
cat gpurun_out/r3n/phase_cost.txt
B="python bench.py --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --steps 20 --warmup 3"
for rps in 131072 262144 1048576; do
  $B --reads-per-step $rps > gpurun_out/r3n/greedy_$rps.json 2> gpurun_out/r3n/greedy_$rps.err
  $B --reads-per-step $rps --anchors > gpurun_out/r3n/anchors_$rps.json 2> gpurun_out/r3n/anchors_$rps.err
  $B --reads-per-step $rps --exhaustive > gpurun_out/r3n/exh_$rps.json 2> gpurun_out/r3n/exh_$rps.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3n/*_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], [(k["kernel"][:34], k["avg_ms"]) for k in d["roofline"]["kernels_ms"]])
PY
