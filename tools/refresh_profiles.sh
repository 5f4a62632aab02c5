#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): regenerates everything profiles/ is built from into
# gpurun_out/refresh/.  tools/collect_profiles.py then copies the summaries into profiles/.
# Round 6: everything runs the reference's arithmetic (fp32-accurate products: flags = 0, the modules' / Trainer's / bench.py's
# default); the opt-in tolerance mode is profiled beside it where a *_fast16 key of the bench line needs backing.
#   part 1: the default bench + kernel statistics        part 2: configs 4 / 5, head
#   part 3: isolated (cold) kernel statistics + FETCH / WRITE counter passes, exact and fast16      part 4: MFMA-busy counters
set -u
unset VQA_PRECISION
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
PART=${1:-all}
mkdir -p gpurun_out/refresh
R="timeout 400 rocprofv3"
drop_traces() { find gpurun_out/refresh -name '*_trace.csv' -delete; }
want() { { [ "$PART" = all ] && [ "$1" != 3b ]; } || [ "$PART" = "$1" ]; }
if want 1; then
timeout 500 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
$R --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --no-cpu-baseline > $O/bench_stats.log 2>&1
$R --kernel-trace --stats --output-format csv -d $O/roofline -- python3 bench.py --only roofline > $O/roofline.log 2>&1
$R --kernel-trace --stats --output-format csv -d $O/hot -- python3 bench.py --only hot > $O/hot.log 2>&1
$R --kernel-trace --stats --output-format csv -d $O/step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/step.log 2>&1
# the full step at the reference's 448 x 448 images (N = 196): its line and its kernels
timeout 400 python3 bench.py --image-size 448 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/step448.json 2> $O/step448.err
$R --kernel-trace --stats --output-format csv -d $O/step448 -- python3 bench.py --image-size 448 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/step448.log 2>&1
drop_traces
tail -c 300 $O/bench.json
fi
if want 2; then
# config 4 (ResNet 7x7x2048 grid, reduced precision): its own bench line, and the kernels of its hot path
timeout 400 python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --no-cpu-baseline > $O/cfg4_bench.json 2> $O/cfg4_bench.err
$R --kernel-trace --stats --output-format csv -d $O/cfg4_hot -- python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --only hot > $O/cfg4_hot.log 2>&1
# config 5 (frozen BERT-base token embeddings as the word level): its bench line on ONE GPU (the driver runs it at --gpus 4)
timeout 300 python3 bench.py --model attention_bert --no-cpu-baseline --no-extras > $O/cfg5_bench.json 2> $O/cfg5_bench.err
# the answer head alone (tools/probe_head.py: HIP head against the stock modules, and the C-ABI calls by themselves)
$R --kernel-trace --stats --output-format csv -d $O/head -- python3 tools/probe_head.py > $O/head.log 2>&1
timeout 200 python3 tools/probe_head.py > $O/head_unprofiled.log 2>&1
drop_traces
tail -c 300 $O/cfg4_bench.json
fi
if want 3; then
# the dominant kernel ALONE, launches rotating over independent buffer sets exactly as bench.py's roofline legs time it (cold):
# the step's own shape (N = 49: `roofline`), the reference's grid (N = 196: `roofline_reference_grid`), channel-major; exact
# (flags = 0) and, for the *_fast16 keys, the tolerance mode; config 4's shape in the reduced-precision mode
for cfg in "49 512 lm exact 0" "196 512 lm exact 0" "196 512 cm exact 0" "49 512 lm fast 0" "196 512 lm fast 0" "49 2048 lm exact 1"; do
  set -- $cfg
  tag=$1_$2_$3_$4; [ $5 = 1 ] && tag=$1_$2_$3_bf16
  N=$1 D=$2 LAYOUT=$3 PRECISION=$4 OPT=$5 ITERS=98 $R --kernel-trace --stats --output-format csv -d $O/iso_$tag -- python3 tools/probe_fwd_one.py > $O/iso_$tag.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    N=$1 D=$2 LAYOUT=$3 PRECISION=$4 OPT=$5 ITERS=28 $R --pmc $c --output-format csv -d $O/pmc_${c}_$tag -- python3 tools/probe_fwd_one.py > $O/pmc_${c}_$tag.log 2>&1
  done
  P=$4; [ $4 = fast ] && P=fast16; [ $5 = 1 ] && P=bf16
  PRODUCTS=$P COLD=1 python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE_$tag $O/pmc_WRITE_SIZE_$tag 160 $1 26 $2 3 $3 >> $O/pmc_traffic.log 2>&1
  find $O/pmc_FETCH_SIZE_$tag $O/pmc_WRITE_SIZE_$tag -name '*counter_collection.csv' -delete
done
# the co-attention forward + backward alone, one shape and one arithmetic per run (per-kernel averages that do not mix shapes),
# iterations rotating over three input sets; then the same two counters per kernel
for N in 49 196; do
  for P in exact fast; do
    VQA_PRECISION=$P $R --kernel-trace --stats --output-format csv -d $O/fb_${N}_$P -- python3 tools/probe_hot.py $N lm 200 > $O/fb_${N}_$P.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      VQA_PRECISION=$P $R --pmc $c --output-format csv -d $O/pmcb_${c}_${N}_$P -- python3 tools/probe_hot.py $N lm 30 > $O/pmcb_${c}_${N}_$P.log 2>&1
    done
    PP=$P; [ $P = fast ] && PP=fast16
    PRODUCTS=$PP python3 tools/pmc_traffic_bwd.py $O/pmcb_FETCH_SIZE_${N}_$P $O/pmcb_WRITE_SIZE_${N}_$P 160 $N 26 512 3 lm >> $O/pmc_traffic.log 2>&1
    find $O/pmcb_FETCH_SIZE_${N}_$P $O/pmcb_WRITE_SIZE_${N}_$P -name '*counter_collection.csv' -delete
  done
done
drop_traces
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cp profiles/pmc_traffic_backward.json $O/pmc_traffic_backward.json
tail -5 $O/pmc_traffic.log | cut -c1-400
fi
if want 3b; then          # only the forward + backward sequences of part 3 (after a change to tools/probe_hot.py or the backward)
for N in 49 196; do
  for P in exact fast; do
    VQA_PRECISION=$P $R --kernel-trace --stats --output-format csv -d $O/fb_${N}_$P -- python3 tools/probe_hot.py $N lm 200 > $O/fb_${N}_$P.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      VQA_PRECISION=$P $R --pmc $c --output-format csv -d $O/pmcb_${c}_${N}_$P -- python3 tools/probe_hot.py $N lm 30 > $O/pmcb_${c}_${N}_$P.log 2>&1
    done
    PP=$P; [ $P = fast ] && PP=fast16
    PRODUCTS=$PP python3 tools/pmc_traffic_bwd.py $O/pmcb_FETCH_SIZE_${N}_$P $O/pmcb_WRITE_SIZE_${N}_$P 160 $N 26 512 3 lm >> $O/pmc_traffic.log 2>&1
    find $O/pmcb_FETCH_SIZE_${N}_$P $O/pmcb_WRITE_SIZE_${N}_$P -name '*counter_collection.csv' -delete
  done
  $R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_fb_$N -- python3 tools/probe_hot.py $N lm 30 > $O/pmc_fb_$N.log 2>&1
done
drop_traces
cp profiles/pmc_traffic_backward.json $O/pmc_traffic_backward.json
fi
if want 4; then
# (the raw counter CSVs of a whole bench leg are hundreds of MB: each pass is reduced to per-kernel sums on the box, tools/pmc_reduce.py)
N=49 ITERS=28 $R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma -- python3 tools/probe_fwd_one.py > $O/pmc_mfma.log 2>&1
python3 tools/pmc_reduce.py $O/pmc_mfma
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_hot -- python3 bench.py --only hot > $O/pmc_hot.log 2>&1
python3 tools/pmc_reduce.py $O/pmc_hot
for N in 49 196; do
  $R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_fb_$N -- python3 tools/probe_hot.py $N lm 30 > $O/pmc_fb_$N.log 2>&1
  python3 tools/pmc_reduce.py $O/pmc_fb_$N
done
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_cfg4_hot -- python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --only hot > $O/pmc_cfg4_hot.log 2>&1
python3 tools/pmc_reduce.py $O/pmc_cfg4_hot
fi
echo "refresh part $PART done"
