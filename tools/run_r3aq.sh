set -e
mkdir -p gpurun_out/r3aq
run() { tag=$1; shift; timeout -k 10 300 python bench.py --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc "$@" > gpurun_out/r3aq/$tag.json 2> gpurun_out/r3aq/$tag.err || { tail -20 gpurun_out/r3aq/$tag.err; return 0; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3aq/$tag.json')); print('$tag', d['value'], d['ms_per_step'], [(k['kernel'][:28], k['avg_ms']) for k in d['roofline']['kernels_ms']][1])"; }
run chr1_near --workload chr1
run g12m_near --genome 12000000
export BGR_LIB_PATH=$PWD/bgreat_amd/lib_nn/libbgreat_gpu.so
run chr1_nonear --workload chr1
run g12m_nonear --genome 12000000
