set -u
mkdir -p gpurun_out/r4j
run() { tag=$1; shift; python tools/e2e.py --reads 100000000 --check 0 --extra "$*" > gpurun_out/r4j/$tag.json 2> gpurun_out/r4j/$tag.err; python -c "
import json; d=json.load(open('gpurun_out/r4j/$tag.json')); print('%-28s run0 %7.1f run1 %7.1f Mreads/s  user %5.1f sys %5.1f' % ('$tag', d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1']['cpu_user_s'], d['run1']['cpu_sys_s']))"; }
run spin_w2 --set blocking_sync=0 --set workers_per_device=2
run spin_w3 --set blocking_sync=0 --set workers_per_device=3
run block_w2 --set blocking_sync=1 --set workers_per_device=2
run block_w3 --set blocking_sync=1 --set workers_per_device=3
run block_w4 --set blocking_sync=1 --set workers_per_device=4
run block_w6 --set blocking_sync=1 --set workers_per_device=6
python bench.py --reads-per-step 262144 --steps 50 --warmup 5 --no-pmc --no-sub --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 > gpurun_out/r4j/small_launch.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r4j/small_launch.json').read().strip().splitlines()[-1]); print('262144-read launches', d['value'], d['roofline']['kernels_ms'])"
