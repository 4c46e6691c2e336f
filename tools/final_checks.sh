set -e
mkdir -p gpurun_out/fin
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/fin/pytest.log 2>&1 || { tail -20 gpurun_out/fin/pytest.log; exit 1; }
tail -n 2 gpurun_out/fin/pytest.log
for w in "ecoli --exhaustive" "ecoli --anchors"; do
  set -- $w
  python bench.py --workload $1 $2 --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc > gpurun_out/fin/$1$2.json 2> gpurun_out/fin/$1$2.log
  python -c "
import json,sys
d=json.load(open('gpurun_out/fin/$1$2.json')); r=d['roofline']; print('$1 $2', d['value'], r['avg_launch_ms'], [k['avg_ms'] for k in r['kernels_ms']], d['parity_sample'])"
done
