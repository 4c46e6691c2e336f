#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- rocprofv3 passes over bench.py on the GPU box (run via gpurun).
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in its own run (SQ / FETCH_SIZE / WRITE_SIZE+L2).
# Output under gpurun_out/prof_<tag>/ ; copy the summaries worth keeping into profiles/.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-pmc --no-sub --cpu-sample 0 --cpu-sample-exh 0 $*"   # (--no-pmc: bench.py must not start its own rocprofv3 child passes inside a profiled run)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/kt_bench.json" 2> "$OUT/kt.err" || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.err"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq_bench.json" 2> "$OUT/pmc_sq.err" || { echo "pmc sq pass failed"; tail -5 "$OUT/pmc_sq.err"; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq2_bench.json" 2> "$OUT/pmc_sq2.err" || { echo "pmc sq2 pass failed"; tail -5 "$OUT/pmc_sq2.err"; exit 1; }
rocprofv3 --pmc FETCH_SIZE TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch_bench.json" 2> "$OUT/pmc_fetch.err" || { echo "pmc fetch pass failed"; tail -5 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write_bench.json" 2> "$OUT/pmc_write.err" || { echo "pmc write pass failed"; tail -5 "$OUT/pmc_write.err"; exit 1; }
python3 "$ROOT/tools/prof_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
