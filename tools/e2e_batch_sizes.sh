#!/bin/bash
# tools/e2e_batch_sizes.sh [reads=50000000] -- GPU box: end-to-end rate of the CLI's text route by piece size (--batch), FASTA, 16 host threads; one input file,
# three rounds interleaved (the first round also warms the page cache)
N=${1:-50000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/bgr_e2eb_XXXX)
python3 - "$ROOT" "$W" "$N" <<'PY' || exit 1
import sys, os
root, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root)
from tools.synth import Synth
s = Synth(4_600_000, 140, 2, 31, 20261003)
s.write_unitigs(os.path.join(w, "u.fa"))
s.write_reads(os.path.join(w, "r.fa"), 0, n, 150, 2, 77, threads=16)
PY
for round in 0 1 2; do
  for b in 262144 524288 1048576 131072; do
    mkdir -p "$W/run" && cd "$W/run"
    "$ROOT/bgreat_amd/bin/bgreat" -r "$W/r.fa" -k 31 -g "$W/u.fa" -m 2 -t 16 --batch $b --set timing=1 > /dev/null 2> "$W/err.txt" || { echo "FAILED batch $b"; tail -3 "$W/err.txt"; }
    echo "round $round  batch $b  $(grep '^bgreat: mapping' "$W/err.txt" | tail -1)"
    cd /tmp && rm -rf "$W/run"
  done
done
rm -rf "$W"
