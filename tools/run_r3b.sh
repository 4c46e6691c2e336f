set -e
mkdir -p gpurun_out/r3b
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r3b/cal -- $R/tools/ubench/valu_rates > $R/gpurun_out/r3b/cal.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/r3b/cal/**/*counter_collection.csv", recursive=True)[0]
acc=collections.OrderedDict()
for row in csv.DictReader(open(f)):
    k=(row["Kernel_Name"], row["Grid_Size"] if "Grid_Size" in row else "")
    acc.setdefault(k, collections.Counter())[row["Counter_Name"]] += float(row["Counter_Value"])
out=open(R+"/gpurun_out/r3b/cal_summary.txt","w")
for k,c in acc.items():
    n=c["SQ_INSTS_VALU"] or 1
    out.write("%-40s grid %-8s insts %.3e thr_cyc/inst %.2f active/inst %.3f active2/inst %.3f busy %.3e wavecyc %.3e busycu %.3e\n" % (k[0][:40],k[1],n,c["SQ_THREAD_CYCLES_VALU"]/n,c["SQ_ACTIVE_INST_VALU"]/n,c["SQ_ACTIVE_INST_VALU2"]/n,c["SQ_BUSY_CYCLES"],c["SQ_WAVE_CYCLES"],c["SQ_BUSY_CU_CYCLES"]))
out.close()
print(open(R+"/gpurun_out/r3b/cal_summary.txt").read()[:6000])
PY
