set -e
mkdir -p gpurun_out/r3ae
timeout -k 10 900 python -m pytest tests/test_gpu_text.py -x -q > gpurun_out/r3ae/text_tests.log 2>&1 || { tail -40 gpurun_out/r3ae/text_tests.log; exit 1; }
tail -3 gpurun_out/r3ae/text_tests.log
run() { tag=$1; shift; BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 200000 "$@" > gpurun_out/r3ae/$tag.json 2> gpurun_out/r3ae/$tag.err || { tail -20 gpurun_out/r3ae/$tag.err; exit 1; }; python3 -c "
import json; d=json.load(open('gpurun_out/r3ae/$tag.json')); print('$tag', d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1']['cpu_user_s'], d['run1']['cpu_sys_s'], d.get('check'))"; }
run fastq_text --fastq
grep "bgreat:" gpurun_out/r3ae/fastq_text.err | head -8
run fastq_host --fastq --extra=--host-route
