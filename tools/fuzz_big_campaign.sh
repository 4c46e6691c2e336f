# tools/fuzz_big_campaign.sh [seed offset] -- on the GPU box: tools/fuzz_big_batches.py (40 k .. 400 k reads per random configuration, every row against the oracle)
# over greedy / anchors / exhaustive seeds; prints the totals, exit code 1 on any mismatch
O=${1:-0}
mkdir -p gpurun_out/fuzzbig
rc=0
run() { tag=$1; shift; timeout -k 10 420 python tools/fuzz_big_batches.py "$@" > gpurun_out/fuzzbig/$tag.log 2>&1 || rc=1; tail -1 gpurun_out/fuzzbig/$tag.log; grep MISMATCH gpurun_out/fuzzbig/$tag.log | head -3; }
run g1 $((1 + O)) greedy 12
run g2 $((2 + O)) greedy 12
run a3 $((3 + O)) anchors 8
run e4 $((4 + O)) exhaustive 8
echo "rc $rc"
exit $rc
