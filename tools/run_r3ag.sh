set -e
mkdir -p gpurun_out/r3ag
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "minimizer_filter or mphf_in_lds or fallback_list" > gpurun_out/r3ag/tests.log 2>&1 || { tail -40 gpurun_out/r3ag/tests.log; exit 1; }
tail -3 gpurun_out/r3ag/tests.log
timeout -k 10 500 python bench.py --workload chr1 --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 > gpurun_out/r3ag/chr1.json 2> gpurun_out/r3ag/chr1.err || { tail -20 gpurun_out/r3ag/chr1.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3ag/chr1.json')); print(d['value'], d['ms_per_step'], d['roofline'])"
