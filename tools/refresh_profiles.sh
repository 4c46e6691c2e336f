#!/bin/bash
# tools/refresh_profiles.sh <part> -- on the GPU box (gpurun): the rocprofv3 summaries and bench lines kept under profiles/ for this round.
#   part 1: kernel-trace + PMC summaries (tools/prof.sh) of the five workloads        -> gpurun_out/prof_r05_*/summary.txt, kt stats
#   part 2: bench lines with their own PMC passes, and the mode x batch size matrix   -> gpurun_out/r05/*.json
# Copy what should be judged into profiles/ afterwards (tools/collect_profiles.sh).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out/r05
if [ "$1" = 1 ]; then
  bash tools/prof.sh r05_ecoli --e2e-reads 0 --pcie-steps 0 > gpurun_out/prof_r05_ecoli.log 2>&1 || { tail -20 gpurun_out/prof_r05_ecoli.log; exit 1; }
  for w in chr1 branchy small; do
    bash tools/prof.sh r05_$w --e2e-reads 0 --pcie-steps 0 --workload $w > gpurun_out/prof_r05_$w.log 2>&1 || { tail -20 gpurun_out/prof_r05_$w.log; exit 1; }
  done
  bash tools/prof.sh r05_anchors --e2e-reads 0 --pcie-steps 0 --anchors > gpurun_out/prof_r05_anchors.log 2>&1 || { tail -20 gpurun_out/prof_r05_anchors.log; exit 1; }
  for w in ecoli chr1 branchy small anchors; do echo "== $w"; grep -A2 "VALU busy issue slots" gpurun_out/prof_r05_$w/summary.txt | head -8; done
else
  for w in chr1 branchy small; do
    python bench.py --workload $w --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --cpu-sample-exh 0 --no-sub > gpurun_out/r05/bench_$w.json 2> gpurun_out/r05/bench_$w.err || { tail -20 gpurun_out/r05/bench_$w.err; exit 1; }
  done
  python bench.py --anchors --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --cpu-sample-exh 0 --no-sub > gpurun_out/r05/bench_anchors.json 2> gpurun_out/r05/bench_anchors.err
  python bench.py --exhaustive --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --cpu-sample-exh 0 --no-sub > gpurun_out/r05/bench_ecoli_exhaustive.json 2> gpurun_out/r05/bench_ecoli_exhaustive.err
  for mode in greedy anchors exhaustive; do
    for n in 131072 262144 1048576 5000000; do
      extra=""; [ $mode = anchors ] && extra="--anchors"; [ $mode = exhaustive ] && extra="--exhaustive"
      steps=10; [ $n -lt 1000000 ] && steps=50
      python bench.py $extra --reads-per-step $n --steps $steps --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --cpu-sample-exh 0 --no-pmc --no-sub > gpurun_out/r05/matrix_${mode}_$n.json 2> gpurun_out/r05/matrix_${mode}_$n.err || { tail -20 gpurun_out/r05/matrix_${mode}_$n.err; exit 1; }
    done
  done
  python - <<'PY'
import json, glob
for w in ("chr1", "branchy", "small", "anchors", "ecoli_exhaustive"):
    d = json.load(open("gpurun_out/r05/bench_%s.json" % w)); r = d["roofline"]
    print(w, d["value"], r["bound"], r["frac"], "hbm", r["hbm"]["frac"], "traffic_frac", r.get("traffic_frac"), "l2hit", r.get("l2_hit_rate"), "l2req/read", r.get("l2_requests_per_read"), "valu", (r.get("valu_issue") or {}).get("frac"))
for f in sorted(glob.glob("gpurun_out/r05/matrix_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
fi
