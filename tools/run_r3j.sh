mkdir -p gpurun_out/r3j
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tools.synth import Synth
s = Synth(4_600_000, 140, 2, 31, 20261003)
s.write_unitigs("/tmp/u.fa")
s.write_reads("/tmp/r.fa", 0, 30_000_000, 150, 2, 77, threads=16)
PY
mkdir -p /tmp/run && cd /tmp/run
BGREAT_TIMING=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3j/kt -- $R/bgreat_amd/bin/bgreat -r /tmp/r.fa -k 31 -g /tmp/u.fa -m 2 -t 16 --batch 262144 > $R/gpurun_out/r3j/cli.txt 2>&1
grep "bgreat:" $R/gpurun_out/r3j/cli.txt
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/r3j/kt/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
ev=[]
for r in rows:
    nm=r["Kernel_Name"]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    kind="copy" if "copyBuffer" in nm else ("fill" if "fillBuffer" in nm else ("greedy" if "greedy_multi" in nm else ("records" if "text_records" in nm else "other")))
    ev.append((s,e,kind,r.get("Queue_Id","")))
t0=min(s for s,e,k,q in ev); t1=max(e for s,e,k,q in ev)
print("span %.3f ms, kernels %d, queues %s" % ((t1-t0)/1e6, len(ev), sorted(set(q for *_,q in ev))))
def union(iv):
    iv=sorted(iv); tot=0; cs,ce=None,None
    for s,e in iv:
        if cs is None: cs,ce=s,e
        elif s<=ce: ce=max(ce,e)
        else: tot+=ce-cs; cs,ce=s,e
    if cs is not None: tot+=ce-cs
    return tot
for k in ("copy","greedy","records","other","fill"):
    iv=[(s,e) for s,e,kk,q in ev if kk==k]
    print("%-8s n %5d sum %.3f ms union %.3f ms" % (k, len(iv), sum(e-s for s,e in iv)/1e6, union(iv)/1e6))
allu=union([(s,e) for s,e,k,q in ev]); print("all: sum %.3f union %.3f ms" % (sum(e-s for s,e,k,q in ev)/1e6, allu/1e6))
cu=union([(s,e) for s,e,k,q in ev if k=="copy"]); nu=union([(s,e) for s,e,k,q in ev if k!="copy"])
print("copy union %.3f + non-copy union %.3f vs all union %.3f -> overlap %.3f ms" % (cu/1e6, nu/1e6, allu/1e6, (cu+nu-allu)/1e6))
# big copies: durations
big=sorted([(e-s) for s,e,k,q in ev if k=="copy" and e-s>200000])
if big: print("big copies: n %d median %.3f ms min %.3f max %.3f" % (len(big), big[len(big)//2]/1e6, big[0]/1e6, big[-1]/1e6))
PY
