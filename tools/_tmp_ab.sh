set -e
mkdir -p gpurun_out/r3ax
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "greedy or minimizer or golden or random or mphf or fallback" > gpurun_out/r3ax/tests.log 2>&1 || { tail -40 gpurun_out/r3ax/tests.log; exit 1; }
tail -2 gpurun_out/r3ax/tests.log
run() { tag=$1; shift; timeout -k 10 300 python bench.py --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc "$@" > gpurun_out/r3ax/$tag.json 2> gpurun_out/r3ax/$tag.err || { tail -20 gpurun_out/r3ax/$tag.err; return 0; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3ax/$tag.json')); print('$tag', d['value'], d['ms_per_step'], [(k['kernel'][:28], k['avg_ms']) for k in d['roofline']['kernels_ms']][1])"; }
run ecoli
run chr1 --workload chr1
run small --workload small
run g7m --genome 7000000
