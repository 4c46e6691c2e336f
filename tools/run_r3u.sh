set -e
mkdir -p gpurun_out/r3u
python bench.py > gpurun_out/r3u/bench_default.json 2> gpurun_out/r3u/bench_default.err || { tail -20 gpurun_out/r3u/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3u/bench_default.json"))
r=d["roofline"]
print(d["value"], d["ms_per_step"])
print({k:v for k,v in r.items() if k not in ("pmc_per_kernel","hbm","valu_issue","kernels_ms") and "note" not in k and k!="launch"})
print("hbm", {k:v for k,v in r["hbm"].items() if k!="definition"})
print("valu", {k:v for k,v in (r.get("valu_issue") or {}).items() if k!="definition"})
print("cpu", d["cpu_baseline"])
print("e2e", d["e2e"])
print("pcie", d["pcie_inclusive"]["value"], d["pcie_inclusive"]["host_packed"]["value"])
PY
