#!/bin/bash
# Where the four-reads-per-wave greedy kernel spends its instructions: the same launch with the kernel cut short behind the
# staging of the reads (1) and behind the anchor scan (2), next to the full kernel (0); needs the diagnostic build
#   make -C bgreat_amd BUILD=build_phase LIBDIR=lib_phase EXTRA=-DBGR_PHASE_TIMING lib_phase/libbgreat_gpu.so
# usage (GPU box): tools/phase_cost.sh [bench.py workload flags]   -> gpurun_out/phase_cost_<stop>.json
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export BGR_LIB_PATH=$PWD/bgreat_amd/lib_phase/libbgreat_gpu.so
for stop in 0 1 2; do
    python bench.py --steps 3 --warmup 1 --e2e-reads 0 --pcie-steps 0 --cpu-sample 0 --debug-stop $stop "$@" > gpurun_out/phase_cost_$stop.json 2> gpurun_out/phase_cost_$stop.log
    python - "$stop" <<'PY'
import json, sys
d = json.load(open("gpurun_out/phase_cost_%s.json" % sys.argv[1]))
r = d["roofline"]
print("stop", sys.argv[1], "launch %.3f ms" % r["avg_launch_ms"], [(k["kernel"][:40], k["avg_ms"]) for k in r["kernels_ms"][:3]],
      "VALU/read", r.get("valu_insts_per_read"), "SALU/read", r.get("salu_insts_per_read"))
for k, v in (r.get("pmc_per_kernel") or {}).items():
    print("   ", k, {c: round(x / 1e6, 1) for c, x in v.items() if c.startswith("SQ_")})
PY
done
