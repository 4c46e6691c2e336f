set -e
mkdir -p gpurun_out/r3ah
for f in 2 1 0; do
  BGREAT_BLOOM=$f timeout -k 10 300 python bench.py --workload chr1 --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --no-pmc > gpurun_out/r3ah/chr1_f$f.json 2> gpurun_out/r3ah/chr1_f$f.err || { tail -20 gpurun_out/r3ah/chr1_f$f.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3ah/chr1_f$f.json')); print('filter env $f', d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
done
