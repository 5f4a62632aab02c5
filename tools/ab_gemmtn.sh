#!/bin/bash
# Knock-out timing of the backward's GEMM launch (gemm_tn_kernel inside coattn_backward): builds variants of the library
# with -DGEMMTN_KO=<mask> HERE (cross-compile), then `tools/ab_gemmtn.sh run` ON THE GPU BOX times them (wrong results).
#   tools/ab_gemmtn.sh build 1 2 4 8 16 32 ...      tools/ab_gemmtn.sh run 1 2 4 8 16 32 ...
# FILE=gemm_tn_wide MACRO=GEMMTNW_KO: the same for gemm_tn_wide_kernel (run with COATTN_NO_COMBINE=1 to time the weight
# gradients without the dQ projection's tiles)
FILE=${FILE:-gemm_tn}; MACRO=${MACRO:-GEMMTN_KO}
mode=$1; shift
if [ "$mode" = build ]; then
  cd "$(dirname "$0")/../visual-question-answering_amd/csrc"
  for ko in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000 \
      -D$MACRO=$ko ${EXTRA:-} -c $FILE.hip -o /tmp/${FILE}_ko$ko.o &
  done
  wait
  for ko in "$@"; do
    objs=$(ls *.o | grep -v "^$FILE.o$" | tr "\n" " ")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libcoattn_tnko$ko.so $objs /tmp/${FILE}_ko$ko.o -Wl,-rpath,/opt/rocm/lib
  done
  exit 0
fi
export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for N in ${NS:-196 49}; do
for ko in base "$@"; do
  if [ $ko = base ]; then unset COATTN_LIB_PATH; else export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_tnko$ko.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tnko_$ko -- python3 tools/probe_hot.py $N lm 100 > gpurun_out/tnko_$ko.log 2>&1
  echo "N=$N KO=$ko $(python3 tools/kstats.py gpurun_out/tnko_$ko 20 | grep gemm_tn | sed 's/.*calls/calls/')"
  rm -rf gpurun_out/tnko_$ko
done
done
