# tools/fuzz_campaign.sh -- on the GPU box: tools/fuzz_parity.py over 57 seeds x 60 random configurations (greedy / exhaustive / anchors; with the minimizer filter forced onto
# the small graphs and without), every row against the oracle.  Prints the totals; exit code 1 on any mismatch.
# usage: tools/fuzz_campaign.sh [seed offset]   (some exhaustive seeds -- tiny search caps on branchy graphs -- take 5 minutes each, oracle included:
# offset 0 runs in ~6 minutes, offset 1000 needs more than one 20-minute gpurun call)
O=${1:-0}
mkdir -p gpurun_out/fuzz
rc=0; bad=0; tot=0
run() { tag=$1; shift; timeout -k 10 400 python tools/fuzz_parity.py "$@" > gpurun_out/fuzz/$tag.log 2>&1 || rc=1; b=$(tail -1 gpurun_out/fuzz/$tag.log | sed 's/.*bad \([0-9]*\).*/\1/'); bad=$((bad + b)); tot=$((tot + 60)); grep MISMATCH gpurun_out/fuzz/$tag.log | head -3; }
export BGR_FUZZ_OPTIONS=build_filter=2,exh_filter=1   # (library options: the tools pass them to bgr_set_option)
for s in $(seq 100 124); do run gf_$s $((s + O)) greedy; done
for s in $(seq 200 204); do run ef_$s $((s + O)) exhaustive auto; done
run ef_lv $((210 + O)) exhaustive by-level
run ef_df $((211 + O)) exhaustive depth-first
unset BGR_FUZZ_OPTIONS
for s in $(seq 300 314); do run g_$s $((s + O)) greedy; done
for s in $(seq 400 404); do run a_$s $((s + O)) anchors; done
for s in $(seq 500 504); do run e_$s $((s + O)) exhaustive auto; done
echo "configurations $tot mismatching $bad rc $rc"
exit $rc
