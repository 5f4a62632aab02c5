python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | tail -2 &&
COATTN_GEMM_W=0 python3 bench.py --only hot > gpurun_out/hot_w0.json 2>gpurun_out/hot_w0.err &&
COATTN_GEMM_W=1 python3 bench.py --only hot > gpurun_out/hot_w1.json 2>gpurun_out/hot_w1.err &&
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hs14 -- python3 bench.py --only hot > gpurun_out/hot.log 2>&1
python3 - <<PY
import csv,glob,json
for t in ("w0","w1"):
    d=json.loads(open("gpurun_out/hot_%s.json"%t).read().strip().splitlines()[-1])
    for k,v in d.items():
        if isinstance(v,dict) and "coattn_fwd_bwd_ms" in v: print(t,k,v["coattn_fwd_bwd_ms"],v["ms_per_step"])
        if isinstance(v,list):
            for e in v:
                if isinstance(e,dict) and "coattn_fwd_bwd_ms" in e: print(t,e.get("N"),e.get("layout"),e["coattn_fwd_bwd_ms"],e["ms_per_step"])
f=sorted(glob.glob("gpurun_out/hs14/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if "gemm" in r["Name"] or "wsplit" in r["Name"]: print("%-70s calls %s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
