set -e
mkdir -p gpurun_out/r3ai
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "minimizer_filter or exhaustive or branchy" > gpurun_out/r3ai/tests.log 2>&1 || { tail -40 gpurun_out/r3ai/tests.log; exit 1; }
tail -3 gpurun_out/r3ai/tests.log
for f in on off; do
  if [ $f = off ]; then export BGREAT_EXH_FILTER_OFF=1; fi
  timeout -k 10 400 python bench.py --workload branchy --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc > gpurun_out/r3ai/branchy_$f.json 2> gpurun_out/r3ai/branchy_$f.err || { tail -20 gpurun_out/r3ai/branchy_$f.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3ai/branchy_$f.json')); print('filter $f', d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
done
