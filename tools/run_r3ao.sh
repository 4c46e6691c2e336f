set -e
mkdir -p gpurun_out/r3ao
run() { tag=$1; shift; timeout -k 10 300 python bench.py --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc "$@" > gpurun_out/r3ao/$tag.json 2> gpurun_out/r3ao/$tag.err || { tail -20 gpurun_out/r3ao/$tag.err; return 0; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3ao/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['launch'], d['config'].get('graph'), [(k['kernel'][:28], k['avg_ms']) for k in d['roofline']['kernels_ms']][1])"; }
run default
run g7m_auto --genome 7000000
run g7m_unstaged --genome 7000000 --lds-mphf 1
run g7m_unstaged_sparse --genome 7000000 --lds-mphf 1 --gamma 1.8
run g10m_auto --genome 10000000
run g10m_unstaged_sparse --genome 10000000 --lds-mphf 1 --gamma 1.8
run g12m_auto --genome 12000000
