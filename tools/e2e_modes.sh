#!/bin/bash
# tools/e2e_modes.sh -- on the GPU box: bin/bgreat end to end (file -> paths + notAligned.fa, separate process per run, page-cache input) per mode and
# route, 100 M x 150 bp reads, 16 host threads; two runs each (tools/e2e.py), first 200 000 records compared with the reference binary's -t 1 bytes.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
O=gpurun_out/r05_e2e
mkdir -p $O
run() { tag=$1; shift; python tools/e2e.py --reads 100000000 --check 200000 "$@" > $O/$tag.json 2> $O/$tag.err || { tail -20 $O/$tag.err; exit 1; }; python3 -c "
import json; d=json.load(open('$O/$tag.json')); print('%-22s run0 %7.1f  run1 %7.1f Mreads/s   cpu user %5.1f s sys %5.1f s   %s' % ('$tag', d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1']['cpu_user_s'], d['run1']['cpu_sys_s'], d.get('check')))"; }
run greedy_text
run greedy_host --extra=--host-route
run correction_text --extra=-c
run correction_host "--extra=-c --host-route"
run anchors_text --extra=-G
run exhaustive_counts_text --extra=-b --check 0
run exhaustive_counts_host "--extra=-b --host-route" --check 0
run exhaustive_write_text "--extra=-b --write-exhaustive" --check 0
run exhaustive_write_host "--extra=-b --write-exhaustive --host-route" --check 0
run fastq_text --fastq
run fastq_host --fastq --extra=--host-route
