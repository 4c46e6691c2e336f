set -e
mkdir -p gpurun_out/r3an
run() { tag=$1; shift; timeout -k 10 300 python bench.py --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc "$@" > gpurun_out/r3an/$tag.json 2> gpurun_out/r3an/$tag.err || { tail -20 gpurun_out/r3an/$tag.err; return 0; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3an/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['launch'], [(k['kernel'][:28], k['avg_ms']) for k in d['roofline']['kernels_ms']][1])"; }
run default
run gamma103_2x16 --gamma 1.03 --waves 16 --blocks-per-cu 2 --lds-mphf 2
run gamma103_2x14 --gamma 1.03 --waves 14 --blocks-per-cu 2 --lds-mphf 2
run gamma100_2x16 --gamma 1.0 --waves 16 --blocks-per-cu 2 --lds-mphf 2
run w14 --waves 14 --blocks-per-cu 2 --lds-mphf 2
run w13 --waves 13 --blocks-per-cu 2 --lds-mphf 2
run w8x3 --waves 8 --blocks-per-cu 3 --lds-mphf 2
