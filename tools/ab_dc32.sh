#!/bin/bash
# Knock-out timing of bwd_dc32_kernel (coattn_bwd32.hip) through tools/probe_marks.py: `build` HERE compiles variants with
# -DDC32_KO=<mask> (wrong results: 1 tanh' arithmetic, 2 operand splits, 4 MFMAs, 8 P_v fragment loads), `run` ON THE GPU BOX.
#   tools/ab_dc32.sh build 1 2 4 8 3 15      tools/ab_dc32.sh run 1 2 4 8 3 15      (NAME=VALUE instead of a mask: that define)
mode=$1; shift
if [ "$mode" = build ]; then
  cd "$(dirname "$0")/../visual-question-answering_amd/csrc"
  mkdir -p ../../tools/ab
  for ko in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000 \
      $(case $ko in *=*) echo -D$ko;; *) echo -DDC32_KO=$ko;; esac) -c coattn_bwd32.hip -o /tmp/coattn_bwd32_$ko.o &
  done
  wait
  for ko in "$@"; do
    objs=$(ls *.o | grep -v "^coattn_bwd32.o$" | tr "\n" " ")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libcoattn_dc_$ko.so $objs /tmp/coattn_bwd32_$ko.o -Wl,-rpath,/opt/rocm/lib
  done
  exit 0
fi
cd $GRAFT_REPO_ROOT
for ko in base "$@"; do
  if [ $ko = base ]; then unset COATTN_LIB_PATH; else export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_dc_$ko.so; fi
  for N in 196 49; do echo "== $ko: $(python3 tools/probe_marks.py $N 2>&1 | grep 'N=' | sed 's/.*bwd_pre [0-9.]* //; s/ bwd_nat32.*//')  (N=$N)"; done
done
