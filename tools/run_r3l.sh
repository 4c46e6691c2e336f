set -e
mkdir -p gpurun_out/r3l
timeout -k 10 600 python -m pytest tests/test_gpu_text.py -x -q -m gpu > gpurun_out/r3l/pytest_text.txt 2>&1 || { tail -40 gpurun_out/r3l/pytest_text.txt; exit 1; }
tail -2 gpurun_out/r3l/pytest_text.txt
for w in 2 3 4; do for b in 262144; do
BGREAT_WORKERS_PER_DEVICE=$w BGREAT_TIMING=1 python tools/e2e.py --reads 100000000 --check 300000 --batch $b > gpurun_out/r3l/e2e_w${w}_b$b.json 2> gpurun_out/r3l/e2e_w${w}_b$b.err || { tail -20 gpurun_out/r3l/e2e_w${w}_b$b.err; exit 1; }
echo "workers $w batch $b"; grep "bgreat:" gpurun_out/r3l/e2e_w${w}_b$b.err | tail -5; python3 -c "
import json; d=json.load(open('gpurun_out/r3l/e2e_w${w}_b$b.json')); print(d['run0']['mreads_per_s'], d['run1']['mreads_per_s'], d['run1'], d.get('check'))"
done; done
