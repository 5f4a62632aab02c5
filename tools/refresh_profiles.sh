#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): regenerates everything profiles/ is built from into
# gpurun_out/refresh/.  tools/collect_profiles.py then copies the summaries into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
PART=${1:-all}                                    # 1: the default bench + kernel statistics; 2: config 4, head, counters
mkdir -p $O
if [ $PART != 2 ]; then
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --no-cpu-baseline > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python3 bench.py --only roofline > $O/roofline.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hot -- python3 bench.py --only hot > $O/hot.log 2>&1
# the headline kernel alone (cfg-2 shape, location-major), timed exactly as bench.py's roofline leg times it
ITERS=100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline -- python3 tools/probe_fwd_one.py > $O/headline.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/step.log 2>&1
rm -f $O/*/*/*kernel_trace.csv $O/hot/*/*_trace.csv
tail -c 300 $O/bench.json
fi
if [ $PART = 1 ]; then exit 0; fi
# config 4 (ResNet 7x7x2048 grid, reduced precision): its own bench line, and the kernels of its hot path
python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --no-cpu-baseline > $O/cfg4_bench.json 2> $O/cfg4_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4_hot -- python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --only hot > $O/cfg4_hot.log 2>&1
# the answer head alone (tools/probe_head.py: HIP head against the stock modules, and the C-ABI calls by themselves)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head -- python3 tools/probe_head.py > $O/head.log 2>&1
python3 tools/probe_head.py > $O/head_unprofiled.log 2>&1
rm -f $O/cfg4_hot/*/*_trace.csv $O/head/*/*_trace.csv
# counter passes of the headline kernel (nothing else in the process): the cfg-2 shape in both layouts, the train step's
# own grid (N = 49), and config 4's shape; FETCH_SIZE and WRITE_SIZE in separate runs
for cfg in "196 512 lm" "196 512 cm" "49 512 lm" "49 2048 lm"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    N=$1 D=$2 LAYOUT=$3 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$1_$2_$3 -- python3 tools/probe_fwd_one.py > $O/pmc_${c}_$1_$2_$3.log 2>&1
  done
  python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE_$1_$2_$3 $O/pmc_WRITE_SIZE_$1_$2_$3 160 $1 26 $2 3 $3 >> $O/pmc_traffic.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma -- python3 tools/probe_fwd_one.py > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_hot -- python3 bench.py --only hot > $O/pmc_hot.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_cfg4_hot -- python3 bench.py --model attention_resnet --opt-lvl 1 --num-cls 3000 --only hot > $O/pmc_cfg4_hot.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
tail -c 300 $O/cfg4_bench.json
