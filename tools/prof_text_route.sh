#!/bin/bash
# tools/prof_text_route.sh [reads=20000000] -- on the GPU box (via gpurun): rocprofv3 kernel + memory-copy trace of the CLI's default
# end-to-end route (bin/bgreat: FASTA text in, record bytes out; then the same with -q), summarised per kernel (calls, mean us, bytes
# it streams per call and the HBM GB/s that is) and per copy direction (GB/s against the link).  Output: gpurun_out/prof_text/summary.txt
set -u
N=${1:-20000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_text
mkdir -p "$OUT"
W=$(mktemp -d /tmp/bgr_prof_text_XXXX)
cd /tmp && export TMPDIR=/tmp
python3 - "$ROOT" "$W" "$N" <<'PY' || exit 1
import sys, os
root, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root)
from tools.synth import Synth
s = Synth(4_600_000, 140, 2, 31, 20261003)
s.write_unitigs(os.path.join(w, "u.fa"))
s.write_reads(os.path.join(w, "r.fa"), 0, n, 150, 2, 77, threads=16)
s.write_reads(os.path.join(w, "r.fq"), 0, n, 150, 2, 77, fastq=True, threads=16)
PY
for mode in fasta fastq; do
  if [ $mode = fasta ]; then ARGS="-r $W/r.fa"; else ARGS="-r $W/r.fq -q"; fi
  mkdir -p "$W/run_$mode" && cd "$W/run_$mode"
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/$mode" -- "$ROOT/bgreat_amd/bin/bgreat" $ARGS --set timing=1 -k 31 -g "$W/u.fa" -m 2 -t 16 > "$OUT/$mode.stdout" 2> "$OUT/$mode.err" \
     || { echo "profiled run ($mode) failed"; tail -5 "$OUT/$mode.err"; rm -rf "$W"; exit 1; }
  ls -l paths notAligned.fa "$W/r.fa" "$W/r.fq" > "$OUT/$mode.files"
done
cd /tmp
python3 "$ROOT/tools/prof_text_summary.py" "$OUT" "$N" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
rm -rf "$W"
