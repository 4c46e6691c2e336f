#!/bin/bash
# tools/quick_bench.sh [tag] -- GPU box: the device-resident leg of the five bench workloads without the optional legs (A/B of kernel changes): one line per workload
T=${1:-q}
mkdir -p gpurun_out/qb
for w in ecoli small chr1 branchy; do
  timeout -k 10 300 python bench.py --workload $w --no-sub --no-pmc --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --cpu-sample-exh 0 --steps 20 --warmup 3 > gpurun_out/qb/${T}_$w.json 2> gpurun_out/qb/${T}_$w.err || echo "FAILED $w"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/qb/${T}_$w.json").read().strip().splitlines()[-1])
    print("%-8s %8.1f Mreads/s  %7.4f ms/step  parity %s  kernels %s" % ("$w", d["value"], d["ms_per_step"], d["parity_sample"]["gpu_equals_oracle"], [(k["kernel"].split(" ")[0][-28:], k["avg_ms"]) for k in d["roofline"]["kernels_ms"]]))
except Exception as ex:
    print("$w: no line (%s)" % ex)
PY
done
