# tools/e2e_fastq_matrix.sh -- GPU box: -q end to end (50 M x 150 bp, tools/e2e.py, fresh process per run) over the host-side knobs of the FASTQ text route
set -u
mkdir -p gpurun_out/r4k
run() { tag=$1; shift; extra=""; [ -n "${CHUNK:-}" ] && extra="--chunk-bytes $CHUNK"; python tools/e2e.py --reads 50000000 --fastq --check 0 $extra --extra "$*" > gpurun_out/r4k/$tag.json 2> gpurun_out/r4k/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/r4k/$tag.json')); print('%-28s run0 %7.1f run1 %7.1f Mreads/s  user %5.1f sys %5.1f' % ('$tag', d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1']['cpu_user_s'], d['run1']['cpu_sys_s']))"; grep "stage busy\|pool CPU" gpurun_out/r4k/$tag.err | tail -2; }
run spin --set blocking_sync=0
run block --set blocking_sync=1
CHUNK=8388608 run block_chunk8M --set blocking_sync=1
CHUNK=1048576 run block_chunk1M --set blocking_sync=1
run nogather --set fastq_gather=0 --set blocking_sync=0
