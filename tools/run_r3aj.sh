set -e
mkdir -p gpurun_out/r3aj
for v in g8 g4; do
  if [ $v = g4 ]; then export BGR_LIB_PATH=$PWD/bgreat_amd/lib_g4/libbgreat_gpu.so; fi
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "greedy or minimizer or golden or random or mphf or full_size" > gpurun_out/r3aj/tests_$v.log 2>&1 || { tail -40 gpurun_out/r3aj/tests_$v.log; exit 1; }
  tail -2 gpurun_out/r3aj/tests_$v.log
  for w in ecoli chr1 small; do
    timeout -k 10 300 python bench.py --workload $w --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc > gpurun_out/r3aj/${w}_$v.json 2> gpurun_out/r3aj/${w}_$v.err || { tail -20 gpurun_out/r3aj/${w}_$v.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/r3aj/${w}_$v.json')); print('$v $w', d['value'], d['ms_per_step'], [(k['kernel'][:28], k['avg_ms']) for k in d['roofline']['kernels_ms']])"
  done
done
