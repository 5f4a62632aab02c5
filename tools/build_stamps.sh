#!/bin/bash
# Diagnostic build with in-kernel stamps (never the shipped library): tools/ab/libcoattn_stamps$SUFFIX.so
# usage: tools/build_stamps.sh [suffix] [extra -D flags...]
set -e
SUF="$1"; shift || true
cd "$(dirname "$0")/../visual-question-answering_amd/csrc"
O=../../tools/ab/obj$SUF
mkdir -p $O
for f in api gemm gemm_w gemm_h2 gemm_bf gemm_tn gemm_tn_wide small_kernels coattn_fused coattn_fwd32 coattn_fused_bwd coattn_bwd32 phrase ce head p2p; do
  X=""; { [ $f = coattn_fwd32 ] || [ $f = coattn_bwd32 ] || [ $f = gemm_w ] || [ $f = gemm_h2 ] || [ $f = gemm_tn ] || [ $f = gemm_tn_wide ]; } && X="-fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $X -DCOATTN_STAMPS=1 "$@" -c $f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libcoattn_stamps$SUF.so $O/*.o -Wl,-rpath,/opt/rocm/lib
