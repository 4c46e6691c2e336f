#!/bin/bash
# tools/e2e_options.sh [reads=50000000] -- GPU box: end-to-end rate of the CLI (FASTA, 16 host threads) under pipeline options, one input file, four rounds interleaved
N=${1:-50000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$(mktemp -d /tmp/bgr_e2eo_XXXX)
python3 - "$ROOT" "$W" "$N" <<'PY' || exit 1
import sys, os
root, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root)
from tools.synth import Synth
s = Synth(4_600_000, 140, 2, 31, 20261003)
s.write_unitigs(os.path.join(w, "u.fa"))
s.write_reads(os.path.join(w, "r.fa"), 0, n, 150, 2, 77, threads=16)
PY
for round in 0 1 2 3; do
  for o in "" "--set workers_per_device=3" "--set workers_per_device=4 --set extra_sets=6" "--set extra_sets=6" "--set blocking_sync=1" "-t 12" "-t 24"; do
    mkdir -p "$W/run" && cd "$W/run"
    T="-t 16"; case "$o" in *-t*) T="";; esac
    "$ROOT/bgreat_amd/bin/bgreat" -r "$W/r.fa" -k 31 -g "$W/u.fa" -m 2 $T $o --set timing=1 > /dev/null 2> "$W/err.txt" || { echo "FAILED [$o]"; tail -3 "$W/err.txt"; }
    echo "round $round  [$o]  $(grep '^bgreat: mapping' "$W/err.txt" | tail -1 | sed 's/bgreat: mapping //')"
    cd /tmp && rm -rf "$W/run"
  done
done
rm -rf "$W"
