#!/bin/bash
# tools/stall_census.sh <tag> [bench args...] -- on the GPU box (via gpurun): where do the waves of the mapping kernels spend their cycles?
# Three rocprofv3 --pmc passes (8 SQ counters each) over a 3-launch child run of bench.py; per kernel: wave-cycles split into
# issuing / waiting (any, on s_waitcnt, on LDS), busy cycles per instruction class, instructions in flight (latency x rate), instruction
# fetches, branches, LDS bank conflicts and the share of lanes a vector instruction has switched on.  Output: gpurun_out/census_<tag>/summary.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/census_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--pmc-child --steps 3 --warmup 0 $*"
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"
P4="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p$i.out" 2> "$OUT/p$i.err" || { echo "pass $i failed"; tail -5 "$OUT/p$i.err"; exit 1; }
done
python3 "$ROOT/tools/stall_census.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
