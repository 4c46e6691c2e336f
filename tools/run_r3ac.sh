set -e
mkdir -p gpurun_out/r3ac
timeout -k 10 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -m gpu -k "scale or chr1 or branchy or lds or staged or table or fuzz or random" > gpurun_out/r3ac/pytest.txt 2>&1 || { tail -30 gpurun_out/r3ac/pytest.txt; exit 1; }
tail -2 gpurun_out/r3ac/pytest.txt
B="python bench.py --e2e-reads 0 --pcie-steps 0 --cpu-sample 0"
for bl in 1 0; do
BGREAT_BLOOM=$bl $B --workload chr1 > gpurun_out/r3ac/chr1_b$bl.json 2> gpurun_out/r3ac/chr1_b$bl.err
BGREAT_BLOOM=$bl $B --workload branchy > gpurun_out/r3ac/branchy_b$bl.json 2> gpurun_out/r3ac/branchy_b$bl.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3ac/*.json")):
    d=json.load(open(f)); r=d["roofline"]; print(f.split("/")[-1], d["value"], d["ms_per_step"], "traffic_frac", r.get("traffic_frac"), "B/read", r.get("traffic_bytes_per_read"), "l2hit", r.get("l2_hit_rate"), "l2req/read", r.get("l2_requests_per_read"), "valu", (r.get("valu_issue") or {}).get("frac"), d["parity_sample"]["gpu_equals_oracle"])
PY
