#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): regenerates everything profiles/ is built from into
# gpurun_out/refresh/.  tools/collect_profiles.py then copies the summaries into profiles/.
set -u
export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
PART=${1:-all}                                    # 1: the default bench + kernel statistics; 2: configs 4 / 5, head, counters
mkdir -p gpurun_out/refresh
R="timeout 400 rocprofv3"
drop_traces() { find gpurun_out/refresh -name '*_trace.csv' -delete; }
if [ $PART != 2 ]; then
timeout 400 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
$R --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --no-cpu-baseline > $O/bench_stats.log 2>&1
$R --kernel-trace --stats --output-format csv -d $O/roofline -- python3 bench.py --only roofline > $O/roofline.log 2>&1
$R --kernel-trace --stats --output-format csv -d $O/hot -- python3 bench.py --only hot > $O/hot.log 2>&1
# the dominant kernel alone, timed exactly as bench.py's roofline legs time it: at the step's own shape (N = 49: `roofline`)
# and at the reference's grid (N = 196: `roofline_reference_grid`)
N=49 ITERS=100 $R --kernel-trace --stats --output-format csv -d $O/headline -- python3 tools/probe_fwd_one.py > $O/headline.log 2>&1
N=196 ITERS=100 $R --kernel-trace --stats --output-format csv -d $O/headline196 -- python3 tools/probe_fwd_one.py > $O/headline196.log 2>&1
# the co-attention forward + backward alone, one shape per run (per-kernel averages that do not mix shapes)
for N in 49 196; do
  $R --kernel-trace --stats --output-format csv -d $O/fb_$N -- python3 tools/probe_hot.py $N lm 200 > $O/fb_$N.log 2>&1
done
$R --kernel-trace --stats --output-format csv -d $O/step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/step.log 2>&1
# the full step at the reference's 448 x 448 images (N = 196): its line and its kernels
timeout 400 python3 bench.py --image-size 448 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/step448.json 2> $O/step448.err
$R --kernel-trace --stats --output-format csv -d $O/step448 -- python3 bench.py --image-size 448 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/step448.log 2>&1
drop_traces
tail -c 300 $O/bench.json
fi
if [ $PART = 1 ]; then exit 0; fi
# config 4 (ResNet 7x7x2048 grid, reduced precision): its own bench line, and the kernels of its hot path
timeout 400 python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --no-cpu-baseline > $O/cfg4_bench.json 2> $O/cfg4_bench.err
$R --kernel-trace --stats --output-format csv -d $O/cfg4_hot -- python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --only hot > $O/cfg4_hot.log 2>&1
# config 5 (frozen BERT-base token embeddings as the word level): its bench line on ONE GPU (the driver runs it at --gpus 4)
timeout 300 python3 bench.py --model attention_bert --no-cpu-baseline --no-extras > $O/cfg5_bench.json 2> $O/cfg5_bench.err
# the answer head alone (tools/probe_head.py: HIP head against the stock modules, and the C-ABI calls by themselves)
$R --kernel-trace --stats --output-format csv -d $O/head -- python3 tools/probe_head.py > $O/head.log 2>&1
timeout 200 python3 tools/probe_head.py > $O/head_unprofiled.log 2>&1
drop_traces
# counter passes of the forward kernel (nothing else in the process): the step's own grid (N = 49), the cfg-2 reference shape in
# both layouts, and config 4's shape; FETCH_SIZE and WRITE_SIZE in separate runs
for cfg in "49 512 lm" "196 512 lm" "196 512 cm" "49 2048 lm"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    N=$1 D=$2 LAYOUT=$3 $R --pmc $c --output-format csv -d $O/pmc_${c}_$1_$2_$3 -- python3 tools/probe_fwd_one.py > $O/pmc_${c}_$1_$2_$3.log 2>&1
  done
  python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE_$1_$2_$3 $O/pmc_WRITE_SIZE_$1_$2_$3 160 $1 26 $2 3 $3 >> $O/pmc_traffic.log 2>&1
done
# the same two counters over the co-attention forward + backward, per kernel of the backward (both grids)
for N in 49 196; do
  for c in FETCH_SIZE WRITE_SIZE; do
    $R --pmc $c --output-format csv -d $O/pmcb_${c}_$N -- python3 tools/probe_hot.py $N lm 30 > $O/pmcb_${c}_$N.log 2>&1
  done
  python3 tools/pmc_traffic_bwd.py $O/pmcb_FETCH_SIZE_$N $O/pmcb_WRITE_SIZE_$N 160 $N 26 512 3 lm >> $O/pmc_traffic.log 2>&1
done
N=49 $R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma -- python3 tools/probe_fwd_one.py > $O/pmc_mfma.log 2>&1
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_hot -- python3 bench.py --only hot > $O/pmc_hot.log 2>&1
for N in 49 196; do
  $R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_fb_$N -- python3 tools/probe_hot.py $N lm 30 > $O/pmc_fb_$N.log 2>&1
done
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_cfg4_hot -- python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --only hot > $O/pmc_cfg4_hot.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cp profiles/pmc_traffic_backward.json $O/pmc_traffic_backward.json
tail -c 300 $O/cfg4_bench.json
