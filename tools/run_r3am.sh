set -e
mkdir -p gpurun_out/r3am
timeout -k 10 900 python bench.py > gpurun_out/r3am/bench_default.json 2> gpurun_out/r3am/bench_default.err || { tail -30 gpurun_out/r3am/bench_default.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3am/bench_default.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['bound'], r['frac'], r['valu_issue'], r['kernels_ms'], d['cpu_baseline'], {k:v for k,v in d['config'].items() if 'e2e' in k or 'pcie' in k})"
timeout -k 10 300 python bench.py --reads-per-step 131072 --steps 50 --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc > gpurun_out/r3am/bench_131k.json 2> gpurun_out/r3am/bench_131k.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3am/bench_131k.json')); print('131k', d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
