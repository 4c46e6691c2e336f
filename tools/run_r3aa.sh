set -e
for w in chr1 branchy small; do
bash tools/prof.sh r03_$w --e2e-reads 0 --pcie-steps 0 --workload $w > gpurun_out/prof_r03_$w.log 2>&1 || { tail -20 gpurun_out/prof_r03_$w.log; exit 1; }
done
bash tools/prof.sh r03_anchors --e2e-reads 0 --pcie-steps 0 --anchors > gpurun_out/prof_r03_anchors.log 2>&1 || { tail -20 gpurun_out/prof_r03_anchors.log; exit 1; }
for w in chr1 branchy small anchors; do grep -A3 "VALU busy issue slots" gpurun_out/prof_r03_$w/summary.txt | grep -B1 -A2 "e+0[89]" | head -12; done
python bench.py --workload chr1 --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 > gpurun_out/bench_r03_chr1.json 2> gpurun_out/bench_r03_chr1.err
python bench.py --workload branchy --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 > gpurun_out/bench_r03_branchy.json 2> gpurun_out/bench_r03_branchy.err
python bench.py --workload small --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 > gpurun_out/bench_r03_small.json 2> gpurun_out/bench_r03_small.err
python bench.py --anchors --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 > gpurun_out/bench_r03_anchors.json 2> gpurun_out/bench_r03_anchors.err
python - <<'PY'
import json
for w in ("chr1","branchy","small","anchors"):
    d=json.load(open("gpurun_out/bench_r03_%s.json"%w)); r=d["roofline"]
    print(w, d["value"], r["bound"], r["frac"], "hbm", r["hbm"]["frac"], "traffic_frac", r.get("traffic_frac"), "l2hit", r.get("l2_hit_rate"), "l2req/read", r.get("l2_requests_per_read"), "valu", (r.get("valu_issue") or {}).get("frac"))
PY
