set -e
mkdir -p gpurun_out/r3ak
run() { tag=$1; shift; timeout -k 10 300 python bench.py --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc "$@" > gpurun_out/r3ak/$tag.json 2> gpurun_out/r3ak/$tag.err || { tail -20 gpurun_out/r3ak/$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3ak/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['launch'], [(k['kernel'][:28], k['avg_ms']) for k in d['roofline']['kernels_ms']])"; }
for v in g8 g4; do
  if [ $v = g4 ]; then export BGR_LIB_PATH=$PWD/bgreat_amd/lib_g4/libbgreat_gpu.so; fi
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "greedy or minimizer or golden or random or mphf or full_size or batch" > gpurun_out/r3ak/tests_$v.log 2>&1 || { tail -40 gpurun_out/r3ak/tests_$v.log; exit 1; }
  tail -2 gpurun_out/r3ak/tests_$v.log
  run ecoli_$v --workload ecoli
  run chr1_$v --workload chr1
  run small_$v --workload small
done
run ecoli_g4_staged12 --workload ecoli --waves 12 --blocks-per-cu 2 --lds-mphf 2
run ecoli_g4_staged16 --workload ecoli --waves 16 --blocks-per-cu 1 --lds-mphf 2
