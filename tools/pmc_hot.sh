#!/bin/bash
# Runs ON THE GPU BOX: counter passes over the co-attention forward + backward at one shape (tools/probe_hot.py).
export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
N=${1:-196}
O=gpurun_out/pmc_hot_$N
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 tools/probe_hot.py $N lm 30 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/p2 -- python3 tools/probe_hot.py $N lm 30 > $O/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -- python3 tools/probe_hot.py $N lm 30 > $O/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p4 -- python3 tools/probe_hot.py $N lm 30 > $O/p4.log 2>&1
python3 tools/pmc_kernels.py $O/p1 $O/p2 $O/p3 $O/p4 > $O/summary.txt 2>&1
tail -2 $O/p1.log $O/p2.log | cut -c1-200
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $O/summary.txt
