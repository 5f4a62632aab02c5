# per-variant P_v / P_q kernel time by rocprofv3 kernel trace (developer script)
export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ $v = base ]; then unset COATTN_LIB_PATH; else export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_stamps_$v.so; fi
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gw_$v -- python3 tools/probe_proj.py > gpurun_out/gw_$v.log 2>&1
  python3 - $v <<PY
import csv,glob,collections,sys
f=sorted(glob.glob("gpurun_out/gw_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True))[-1]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "gemm_w" in n:
        agg[(n[:60], r["Grid_Size_X"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items()):
    v2=sorted(v[len(v)//2:]); print("%-8s %-60s grid %8s n %4d med %7.1f min %7.1f" % (sys.argv[1],k[0],k[1],len(v2),v2[len(v2)//2],v2[0]))
PY
done
