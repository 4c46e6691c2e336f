set -e
mkdir -p gpurun_out/r3ap
timeout -k 10 500 python bench.py --workload chr1 --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 > gpurun_out/r3ap/chr1.json 2> gpurun_out/r3ap/chr1.err || { tail -20 gpurun_out/r3ap/chr1.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3ap/chr1.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['bound'], r['frac'], r['valu_issue'], 'l2hit', r['l2_hit_rate'], 'l2req/read', r['l2_requests_per_read'], 'traffic/read', r['traffic_bytes_per_read'], r['kernels_ms'])"
