#!/bin/bash
# Knock-out timing of the pre-split-weight GEMM at the two-piece width (tools/probe_linear.py): `build` HERE compiles variants of
# gemm_w.hip with the developer switches of gemm_w_body.h (wrong results), `run` ON THE GPU BOX times them.
#   tools/ab_gemmw2.sh build NOB NOSPLIT NOMFMA NOA NOSTORE     tools/ab_gemmw2.sh run NOB ...
mode=$1; shift
if [ "$mode" = build ]; then
  cd "$(dirname "$0")/../visual-question-answering_amd/csrc"
  for ko in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000 \
      -DGEMMW_$ko -c gemm_w.hip -o /tmp/gemm_w_$ko.o &
  done
  wait
  for ko in "$@"; do
    objs=$(ls *.o | grep -v "^gemm_w.o$" | tr "\n" " ")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libcoattn_gw$ko.so $objs /tmp/gemm_w_$ko.o -Wl,-rpath,/opt/rocm/lib
  done
  exit 0
fi
cd $GRAFT_REPO_ROOT
for ko in base "$@"; do
  if [ $ko = base ]; then unset COATTN_LIB_PATH; else export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_gw$ko.so; fi
  echo "== $ko"; python3 tools/probe_linear.py 2>&1 | grep -v amdgpu.ids
done
