export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_base.so; else unset COATTN_LIB_PATH; fi
  for N in 196 49; do
    find gpurun_out/abl -name "*_stats.csv" -delete 2>/dev/null
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -- python3 tools/probe_hot.py $N ${LAYOUT:-lm} 200 > gpurun_out/abl.log 2>&1
    python3 - $lib $N <<'PY'
import csv,glob,sys,re
f=sorted(glob.glob("gpurun_out/abl/**/*kernel_stats.csv", recursive=True))[0]
tot=0; parts=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "at::" in n or "rocclr" in n: continue
    k=re.sub(r"void |\(anonymous namespace\)::|[<(].*","",n)
    t=float(r["AverageNs"])/1e3*int(r["Calls"])/300
    tot+=t; parts.append("%s %.1f"%(k.replace("_kernel",""),t))
print(sys.argv[1], "N="+sys.argv[2], "sum %.1f us |"%tot, " ".join(parts))
PY
  done
done
done
