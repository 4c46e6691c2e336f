#!/bin/bash
# tools/ab_pmc.sh TAG [bench flags...] -- GPU box: device-resident leg of one workload WITH the counter passes; prints value, kernel times, VALU / SALU per read, issue fraction
T=$1; shift
mkdir -p gpurun_out/qb
timeout -k 10 400 python bench.py --no-sub --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --cpu-sample-exh 0 --steps 20 --warmup 3 "$@" > gpurun_out/qb/$T.json 2> gpurun_out/qb/$T.err || echo FAILED
python - <<PY
import json
d = json.loads(open("gpurun_out/qb/$T.json").read().strip().splitlines()[-1])
r = d["roofline"]; v = r.get("valu_issue") or {}
print("%-14s %8.1f Mreads/s  kernels %s" % ("$T", d["value"], [(k["kernel"].split(" ")[0][-24:], k["avg_ms"]) for k in r["kernels_ms"]]))
print("               valu/read %s salu/read %s issue frac %s paired %s cycles/inst %s  l2 hit %s traffic B/read %s" % (v.get("valu_insts_per_read"), v.get("salu_insts_per_read"), v.get("frac"), v.get("paired_share"), v.get("cycles_per_valu_inst"), r.get("l2_hit_rate"), r.get("traffic_bytes_per_read")))
pk = r.get("pmc_per_kernel") or {}
for k, c in pk.items():
    print("               %-40s %s" % (k[:40], {x: c[x] for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES") if x in c}))
PY
