#!/bin/bash
# tools/e2e_fastq_gather.sh [reads=30000000] -- GPU box: FASTQ end to end, header + read lines gathered on the host (default) against whole four-line records over the link
N=${1:-30000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/bgr_e2eq_XXXX)
python3 - "$ROOT" "$W" "$N" <<'PY' || exit 1
import sys, os
root, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root)
from tools.synth import Synth
s = Synth(4_600_000, 140, 2, 31, 20261003)
s.write_unitigs(os.path.join(w, "u.fa"))
s.write_reads(os.path.join(w, "r.fq"), 0, n, 150, 2, 77, fastq=True, threads=16)
PY
for round in 0 1 2; do
  for gq in 1 0; do
    mkdir -p "$W/run" && cd "$W/run"
    "$ROOT/bgreat_amd/bin/bgreat" -r "$W/r.fq" -q -k 31 -g "$W/u.fa" -m 2 -t 16 --set timing=1 --set fastq_gather=$gq > /dev/null 2> "$W/err.txt" || { echo "FAILED gather $gq"; tail -3 "$W/err.txt"; }
    echo "round $round  fastq_gather=$gq  $(grep '^bgreat: mapping' "$W/err.txt" | tail -1)  $(sha256sum paths | cut -c1-12)"
    cd /tmp && rm -rf "$W/run"
  done
done
rm -rf "$W"
