# developer script (runs on the GPU box): knock-out builds of gemm_bf.hip timed against the shipped kernel
# usage (local): tools/ab_gemmbf.sh build   -> tools/ab/libcoattn_bf_<ko>.so ; then gpurun -- bash tools/ab_gemmbf.sh run
set -u
KOS=${KOS:-"1 2 3 4 8 16 31"}    # knock-out masks, or NAME=VALUE defines (KOS="BF_EARLY=0" tools/ab_gemmbf.sh build)
if [ "${1:-run}" = build ]; then
  mkdir -p tools/ab
  cd visual-question-answering_amd/csrc
  for k in $KOS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $(case $k in *=*) echo -D$k;; *) echo -DGEMMBF_KO=$k;; esac) -c gemm_bf.hip -o ../../tools/ab/gemm_bf_$k.o
    OBJS=$(ls *.o | grep -v gemm_bf.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libcoattn_bf_$k.so $OBJS ../../tools/ab/gemm_bf_$k.o -Wl,-rpath,/opt/rocm/lib
  done
  exit 0
fi
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for k in 0 $KOS; do
  if [ $k = 0 ]; then unset COATTN_LIB_PATH; else export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_bf_$k.so; fi
  python3 - $k <<'PY'
import sys, os, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from vqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
M, d = 7840, 2048
g = torch.Generator().manual_seed(5)
V = torch.randn(M, d, generator=g).to(dev); W = (torch.randn(d, d, generator=g) / d ** 0.5).to(dev)
bias = torch.zeros(d, device=dev); Y = torch.empty(M, d, device=dev)
wimg = torch.empty(lib.coattn_linear_workspace_bytes(d, d) // 4, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda f: lib.coattn_linear_forward(V.data_ptr(), d, W.data_ptr(), bias.data_ptr(), Y.data_ptr(), wimg.data_ptr(), M, d, d, 0.0, f | 4, st)
call(0)
for _ in range(200): call(1)
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): call(1)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 10)
print("KO %3s: %.1f us per call (windows %s)" % (sys.argv[1], sorted(ts)[1], ["%.1f" % t for t in ts]), flush=True)
PY
done
done
