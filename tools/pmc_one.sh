#!/bin/bash
# Runs ON THE GPU BOX: one counter pass over tools/probe_hot.py: pmc_one.sh N "COUNTER ..." [kernel-name filter]
export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
N=$1; O=gpurun_out/pmc_one
rm -rf $O; mkdir -p $O
rocprofv3 --pmc $2 --output-format csv -d $O/p -- python3 tools/probe_hot.py $N ${LAYOUT:-lm} 30 > $O/p.log 2>&1
python3 tools/pmc_kernels.py $O/p | grep -A2 "${3:-gemm}"
rm -rf $O
