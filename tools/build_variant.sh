#!/bin/bash
# One-file variant of the library for A/B timing (never the shipped one): tools/ab/libcoattn_<suffix>.so
# usage: tools/build_variant.sh <suffix> <file without .hip> [-D flags...]      (the other objects: csrc/*.o of the last build())
set -e
SUF="$1"; F="$2"; shift 2
cd "$(dirname "$0")/../visual-question-answering_amd/csrc"
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000 \
  "$@" -c $F.hip -o /tmp/${F}_$SUF.o
objs=$(ls *.o | grep -v "^$F.o$" | tr "\n" " ")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libcoattn_$SUF.so $objs /tmp/${F}_$SUF.o -Wl,-rpath,/opt/rocm/lib
