set -e
mkdir -p gpurun_out/r3c
python bench.py > gpurun_out/r3c/bench_default.json 2> gpurun_out/r3c/bench_default.err || { tail -20 gpurun_out/r3c/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3c/bench_default.json"))
r=d["roofline"]
print(d["value"], d["ms_per_step"])
print({k:v for k,v in r.items() if k not in ("pmc_per_kernel","hbm","valu_issue","kernels_ms")})
print("hbm", r["hbm"])
print("valu", r.get("valu_issue"))
print("cpu", d["cpu_baseline"])
print("e2e", d["e2e"])
print("pcie", d["pcie_inclusive"])
PY
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3c/pytest.txt 2>&1 || { tail -30 gpurun_out/r3c/pytest.txt; exit 1; }
tail -3 gpurun_out/r3c/pytest.txt
