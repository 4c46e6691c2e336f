# tools/final_checks.sh -- on the GPU box at the end of a round: the GPU test suite, smoke(), and the default bench line (kept as profiles/rNN_bench_ecoli_default.json)
set -e
mkdir -p gpurun_out/fin
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/fin/pytest.log 2>&1 || { tail -20 gpurun_out/fin/pytest.log; exit 1; }
tail -n 2 gpurun_out/fin/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/fin/smoke.log 2>&1 || { tail -20 gpurun_out/fin/smoke.log; exit 1; }
tail -n 1 gpurun_out/fin/smoke.log
T0=$(date +%s)
python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err || { tail -20 gpurun_out/fin/bench_default.err; exit 1; }
echo "bench.py wall: $(( $(date +%s) - T0 )) s"
python -c "
import json
d=json.load(open('gpurun_out/fin/bench_default.json')); r=d['roofline']
print(d['metric'], d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'bound', r['bound'], 'frac', r['frac'], 'hbm frac', r['hbm']['frac'], 'traffic/read', r.get('traffic_bytes_per_read'))
print('kernels', r['kernels_ms'])
print('pcie', d['pcie_inclusive']['value'], d['pcie_inclusive']['two_streams']['value'], d['pcie_inclusive']['host_packed']['value'])
print('e2e', d['e2e']['value'], d['e2e']['runs_mreads_per_s'], d['e2e']['host_route']['value'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['t1']['value'], d['cpu_baseline']['cpu_model'], d['cpu_baseline']['gpu_matches_cpu_records'])
"
