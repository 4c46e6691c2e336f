#!/bin/bash
# tools/collect_profiles.sh -- here (no GPU): copy what tools/refresh_profiles.sh and tools/e2e_modes.sh left under gpurun_out/ into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."

cp gpurun_out/prof_r05_ecoli/summary.txt profiles/r05_ecoli_greedy_5Mx150_summary.txt
cp gpurun_out/prof_r05_chr1/summary.txt profiles/r05_chr1_greedy_5Mx150_summary.txt
cp gpurun_out/prof_r05_branchy/summary.txt profiles/r05_branchy_exhaustive_2Mx250_summary.txt
cp gpurun_out/prof_r05_small/summary.txt profiles/r05_small_greedy_1Mx100_summary.txt
cp gpurun_out/prof_r05_anchors/summary.txt profiles/r05_anchors_5Mx150_summary.txt
cp "$(ls -t gpurun_out/prof_r05_ecoli/kt/runc/*_kernel_stats.csv | head -1)" profiles/r05_ecoli_kernel_stats.csv
for w in chr1 branchy small anchors ecoli_exhaustive; do cp gpurun_out/r05/bench_$w.json profiles/r05_bench_$w.json; done
python3 - <<'PY'
import json, glob
rows = {}
with open("profiles/r05_mode_by_batch_size.jsonl", "w") as out:
    for f in sorted(glob.glob("gpurun_out/r05/matrix_*.json")):
        d = json.load(open(f))
        mode, n = f.split("/")[-1][7:-5].rsplit("_", 1)
        keep = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config")}
        keep["kernels_ms"] = d["roofline"]["kernels_ms"]
        out.write(json.dumps({"mode": mode, "reads_per_launch": int(n), **keep}) + "\n")
        rows.setdefault(mode, {})[int(n)] = (d["value"], d["ms_per_step"])
with open("profiles/r05_mode_by_batch_size.txt", "w") as out:
    out.write("# Mreads/s (ms per launch) of one mapping launch, device-resident reads, E. coli-scale graph, 150 bp, k=31, m=2; bench.py --reads-per-step N [--anchors|--exhaustive] --no-pmc\n")
    out.write("# 131 072 / 262 144 = the batch sizes of the CLI's host route / text route; source lines: r05_mode_by_batch_size.jsonl\n")
    sizes = [131072, 262144, 1048576, 5000000]
    out.write("%-12s" % "mode" + "".join("%22d" % n for n in sizes) + "\n")
    for mode in ("greedy", "anchors", "exhaustive"):
        out.write("%-12s" % mode + "".join("%14.0f (%5.3f)" % rows[mode][n] for n in sizes) + "\n")
print(open("profiles/r05_mode_by_batch_size.txt").read())
PY
python3 tools/kernel_resources.py > profiles/r05_kernel_resources.txt 2>&1
