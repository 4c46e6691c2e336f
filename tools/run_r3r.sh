set -e
mkdir -p gpurun_out/r3r
for w in 2 3; do for b in 131072 262144; do
BGREAT_WORKERS_PER_DEVICE=$w BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 0 --batch $b > gpurun_out/r3r/e2e_w${w}_b$b.json 2> gpurun_out/r3r/e2e_w${w}_b$b.err || { tail -20 gpurun_out/r3r/e2e_w${w}_b$b.err; exit 1; }
echo "workers $w batch $b"; grep "bgreat:" gpurun_out/r3r/e2e_w${w}_b$b.err | head -4; python3 -c "
import json; d=json.load(open('gpurun_out/r3r/e2e_w${w}_b$b.json')); print(d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1'])"
done; done
