#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel statistics of tools/probe_marks.py (coattn_forward + coattn_backward, exact unless MODE=fast)
# on the DEV build for each setting of the developer environment ("-" = defaults); prints the kernels matching GREP.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_dev.so
for setting in "$@"; do
  for N in ${NS:-49}; do
    tag=$(echo "$setting" | tr ' =' '__')
    rm -rf gpurun_out/abk_$tag
    ( [ "$setting" != "-" ] && export $setting; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$tag -- python3 tools/probe_marks.py $N ${LAYOUT:-lm} ${MODE:-exact} > gpurun_out/abk_$tag.log 2>&1 )
    echo "== [$setting] N=$N: $(tail -1 gpurun_out/abk_$tag.log | cut -c1-200)"
    python3 tools/kstats.py gpurun_out/abk_$tag 30 | grep -E "${GREP:-wsplit|gemm_w}"
    rm -rf gpurun_out/abk_$tag
  done
done
