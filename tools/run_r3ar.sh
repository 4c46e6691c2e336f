set -e
mkdir -p gpurun_out/r3ar
run() { tag=$1; shift; timeout -k 10 300 python bench.py --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc "$@" > gpurun_out/r3ar/$tag.json 2> gpurun_out/r3ar/$tag.err || { tail -20 gpurun_out/r3ar/$tag.err; return 0; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3ar/$tag.json')); print('$tag', d['value'], d['ms_per_step'], [(k['kernel'][:28], k['avg_ms']) for k in d['roofline']['kernels_ms']][1])"; }
for g in 12000000 40000000 100000000; do
  run g${g}_filter --genome $g
  BGREAT_BLOOM=0 run g${g}_nofilter --genome $g
done
run chr1 --workload chr1
