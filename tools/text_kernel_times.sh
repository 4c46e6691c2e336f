#!/bin/bash
# tools/text_kernel_times.sh [reads=274000] [reps=8] -- on the GPU box: per-kernel times of bgr_align_fasta_text on one piece (rocprofv3 --kernel-trace --stats around tools/text_bench.py)
N=${1:-274000}; R=${2:-8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/text_kernel_times
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/tools/text_bench.py" "$N" "$R" > "$OUT/stdout.txt" 2>&1 || { tail -5 "$OUT/stdout.txt"; exit 1; }
grep "^rep" "$OUT/stdout.txt" | tail -3
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"].replace("void ", "").replace("bgr::(anonymous namespace)::", "").split("(")[0][:60]
    print("%-60s calls %4s  mean %8.1f us  min %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
