#!/bin/bash
# Knock-out timing of gemm_h2.hip (the tolerance mode's projection kernel) through tools/probe_linear.py: `build` HERE compiles
# variants with -DGEMMH2_KO=<mask> (wrong results: 1 no A reloads, 2 no weight DMA, 4 no MFMAs, 8 no split / LDS writes,
# 16 no C stores) and any extra -D given as NAME=VALUE, `run` ON THE GPU BOX times them.
#   tools/ab_gemmh2.sh build 1 2 4 8 16 31      tools/ab_gemmh2.sh run 1 2 4 8 16 31
mode=$1; shift
if [ "$mode" = build ]; then
  cd "$(dirname "$0")/../visual-question-answering_amd/csrc"
  mkdir -p ../../tools/ab
  for ko in "$@"; do
    case $ko in *=*) def="-D$ko";; *) def="-DGEMMH2_KO=$ko";; esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000 \
      $def -c gemm_h2.hip -o /tmp/gemm_h2_$ko.o &
  done
  wait
  for ko in "$@"; do
    objs=$(ls *.o | grep -v "^gemm_h2.o$" | tr "\n" " ")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libcoattn_h2_$ko.so $objs /tmp/gemm_h2_$ko.o -Wl,-rpath,/opt/rocm/lib
  done
  exit 0
fi
cd $GRAFT_REPO_ROOT
for ko in base "$@"; do
  if [ $ko = base ]; then unset COATTN_LIB_PATH; else export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_h2_$ko.so; fi
  echo "== $ko"; H2ONLY=1 python3 tools/probe_linear.py 2>&1 | grep -v amdgpu.ids
done
