mkdir -p gpurun_out/r3i
env | grep -i -E "^(HSA|ROC|HIP|GPU_|AMD)" | sort > gpurun_out/r3i/env.txt; cat gpurun_out/r3i/env.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default sdma1 blit0; do
  case $v in
    default) ;;
    sdma1) export HSA_ENABLE_SDMA=1 ;;
    blit0) export HSA_ENABLE_SDMA=1; export GPU_FORCE_BLIT_COPY_SIZE=0 ;;
  esac
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/r3i/kt_$v -- python3 $R/tools/text_bench.py 262144 4 > $R/gpurun_out/r3i/tb_$v.txt 2>&1
  echo "== $v"; grep "rep 3" $R/gpurun_out/r3i/tb_$v.txt
  python3 - $v <<'PY'
import csv, glob, os, sys
R=os.environ["GRAFT_REPO_ROOT"]; v=sys.argv[1]
for f in glob.glob(R+"/gpurun_out/r3i/kt_%s/**/*kernel_stats.csv" % v, recursive=True):
    for row in csv.DictReader(open(f)):
        if "copy" in row["Name"].lower() or "fill" in row["Name"].lower(): print("  %-60s calls %4s avg %10.1f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"])/1e3))
for f in glob.glob(R+"/gpurun_out/r3i/kt_%s/**/*memory_copy_stats.csv" % v, recursive=True):
    for row in csv.DictReader(open(f)):
        print("  ", row["Name"], row["Calls"], row["AverageNs"])
PY
done
