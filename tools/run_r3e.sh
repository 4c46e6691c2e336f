set -e
mkdir -p gpurun_out/r3e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/text_bench.py 524288 5 > $R/gpurun_out/r3e/tb.txt 2>&1 || { tail -20 $R/gpurun_out/r3e/tb.txt; exit 1; }
cat $R/gpurun_out/r3e/tb.txt
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/r3e/kt -- python3 $R/tools/text_bench.py 524288 4 > $R/gpurun_out/r3e/tb_prof.txt 2>&1 || { tail -20 $R/gpurun_out/r3e/tb_prof.txt; exit 1; }
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(R+"/gpurun_out/r3e/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-90s calls %4s avg %10.1f us  %5s%%" % (row["Name"][:90], row["Calls"], float(row["AverageNs"])/1e3, row["Percentage"]))
for f in glob.glob(R+"/gpurun_out/r3e/kt/**/*memory_copy_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print(row)
PY
