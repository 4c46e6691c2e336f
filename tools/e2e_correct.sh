#!/bin/bash
# tools/e2e_correct.sh [reads=20000000] -- GPU box: -c (correction mode) end to end with per-kernel device times (rocprofv3 kernel trace) and the CLI's stage timing
N=${1:-20000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/e2e_correct
rm -rf "$OUT"; mkdir -p "$OUT"
W=$(mktemp -d /tmp/bgr_e2ec_XXXX)
cd /tmp && export TMPDIR=/tmp
python3 - "$ROOT" "$W" "$N" <<'PY' || exit 1
import sys, os
root, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root)
from tools.synth import Synth
s = Synth(4_600_000, 140, 2, 31, 20261003)
s.write_unitigs(os.path.join(w, "u.fa"))
s.write_reads(os.path.join(w, "r.fa"), 0, n, 150, 2, 77, threads=16)
PY
mkdir -p "$W/run" && cd "$W/run"
for i in 1 2; do "$ROOT/bgreat_amd/bin/bgreat" -r "$W/r.fa" -k 31 -g "$W/u.fa" -m 2 -t 16 -c --set timing=1 > /dev/null 2> "$OUT/plain_$i.err"; grep "mapping\|text calls\|stage busy\|pool CPU" "$OUT/plain_$i.err"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- "$ROOT/bgreat_amd/bin/bgreat" -r "$W/r.fa" -k 31 -g "$W/u.fa" -m 2 -t 16 -c --set timing=1 > /dev/null 2> "$OUT/prof.err"
F=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    n = r["Name"].replace("void ", "").replace("bgr::(anonymous namespace)::", "").split("(")[0][:56]
    print("%-56s calls %5s  mean %8.1f us  total %7.2f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("all kernels %.1f ms" % (tot / 1e6))
PY
cd /tmp; rm -rf "$W"
