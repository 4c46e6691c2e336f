set -e
mkdir -p gpurun_out/r3k
for numa in 1 0; do for b in 262144; do
BGREAT_NUMA=$numa BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 0 --batch $b > gpurun_out/r3k/e2e_n${numa}_b$b.json 2> gpurun_out/r3k/e2e_n${numa}_b$b.err || { tail -20 gpurun_out/r3k/e2e_n${numa}_b$b.err; exit 1; }
echo "numa $numa batch $b"; grep "bgreat:" gpurun_out/r3k/e2e_n${numa}_b$b.err | tail -5; python3 -c "
import json; d=json.load(open('gpurun_out/r3k/e2e_n${numa}_b$b.json')); print(d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1'])"
done; done
