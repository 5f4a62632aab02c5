#!/bin/bash
# Runs ON THE GPU BOX: the library's own per-launch-group times of coattn_forward + coattn_backward (tools/probe_marks.py,
# exact mode unless MODE=fast) at N = 196 and 49 for each setting of the developer environment given as arguments
# ("COATTN_TN_WIDE3=0" "COATTN_GEMMW_WIDE3=1" ...; "-" = defaults), on the DEV build (make -C csrc DEV=1), twice each.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export COATTN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcoattn_dev.so
for rep in 1 2; do
for setting in "$@"; do
  for N in ${NS:-196 49}; do
    ( [ "$setting" != "-" ] && export $setting; echo "[$setting] $(python3 tools/probe_marks.py $N ${LAYOUT:-lm} ${MODE:-exact} 2>&1 | tail -1)" )
  done
done
done
