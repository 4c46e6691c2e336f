set -e
mkdir -p gpurun_out/r3ad
run() { tag=$1; shift; BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 200000 "$@" > gpurun_out/r3ad/$tag.json 2> gpurun_out/r3ad/$tag.err || { tail -20 gpurun_out/r3ad/$tag.err; exit 1; }; python3 -c "
import json; d=json.load(open('gpurun_out/r3ad/$tag.json')); print('$tag', d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1']['cpu_user_s'], d['run1']['cpu_sys_s'], d.get('check'))"; }
run greedy_text
grep "bgreat:" gpurun_out/r3ad/greedy_text.err | head -5
run greedy_host --extra=--host-route
run correction_text --extra=-c
run correction_host "--extra=-c --host-route"
run anchors_text --extra=-G
run exhaustive_text "--extra=-b --write-exhaustive" --check 0
run exhaustive_counts --extra=-b --check 0
run fastq_host --fastq
