set -e
mkdir -p gpurun_out/r3h
for w in 3 4 6 8; do for b in 131072 262144; do
BGREAT_WORKERS_PER_DEVICE=$w BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 0 --batch $b > gpurun_out/r3h/e2e_w${w}_b$b.json 2> gpurun_out/r3h/e2e_w${w}_b$b.err || { tail -20 gpurun_out/r3h/e2e_w${w}_b$b.err; exit 1; }
echo "workers $w batch $b"; grep "stage busy" gpurun_out/r3h/e2e_w${w}_b$b.err | tail -1; python3 -c "
import json; d=json.load(open('gpurun_out/r3h/e2e_w${w}_b$b.json')); print(d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1'])"
done; done
